// StereoMatcherHIP.cpp — see StereoMatcherHIP.h.  Error behaviour follows the reference: methods
// return void and log with the "[StereoMatcherNew]" prefix (StereoMatcher.cpp:214,253); on a
// back-end failure the previous disparity16 is left untouched (the reference's CUDA back-end calls
// exit() instead, sgbm_cuda/bova_sgm.cu:45-49,79-84).
#include "StereoMatcherHIP.h"

#include <cmath>
#include <iostream>

namespace svh {

StereoMatcherNew::StereoMatcherNew(StereoCamera* s, int dev) : stereo(s), device(dev) {
    sv_default_params(&hip_params, stereo_parameters.numberOfDisparities, paths);
}

StereoMatcherNew::~StereoMatcherNew() {
    if (ctx) sv_destroy(ctx);
}

void StereoMatcherNew::setParameters(int minDisparity, int numberOfDisparities, int SADWindowSize, int disp12MaxDiff,
                                     int preFilterCap, int uniquenessRatio, int speckleWindowSize, int speckleRange,
                                     double sigmaColorBLF, double sigmaSpaceBLF, double wls_lambda, double wls_sigma,
                                     SM_BLF_FILTER BLFfiltering, SM_WLS_FILTER WLSfiltering, SM_MATCHING_ALG stereo_matching) {
    Params& p = stereo_parameters;
    p.minDisparity = minDisparity; p.numberOfDisparities = numberOfDisparities; p.SADWindowSize = SADWindowSize;
    p.disp12MaxDiff = disp12MaxDiff; p.preFilterCap = preFilterCap; p.uniquenessRatio = uniquenessRatio;
    p.speckleWindowSize = speckleWindowSize; p.speckleRange = speckleRange; p.sigmaColorBLF = sigmaColorBLF;
    p.sigmaSpaceBLF = sigmaSpaceBLF; p.wls_lambda = wls_lambda; p.wls_sigma = wls_sigma;
    p.BLFfiltering = BLFfiltering; p.WLSfiltering = WLSfiltering; p.stereo_matching = stereo_matching;
}

// Counterpart of updateCUDAParams (StereoMatcher.cpp:271-289): P1 = 8*BS^2, P2 = 32*BS^2 on gray
// input (no x3, Appendix B-2).  Unlike the CUDA back-end numberOfDisparities reaches the kernels.
void StereoMatcherNew::updateHIPParams() {
    const Params& p = stereo_parameters;
    sv_default_params(&hip_params, p.numberOfDisparities, paths);
    hip_params.min_disparity = p.minDisparity;
    // The GUI / RPC offer SADWindowSize 3..31 in steps of 1 and preFilterCap 0..100 (cvgui.cpp:100,102); the library takes odd
    // blocks up to 15 and caps up to 127 as long as no cost leaves int16 ((2 cap + 63) BS^2 + P2 <= 32767: beyond that the
    // reference's CostType wraps silently).  What lies outside is brought inside here and said so, instead of leaving the
    // module on its previous parameters:
    //  * an even size b: cv::StereoSGBM's window is 2 (b / 2) + 1 = b + 1 wide while matchSGBM's P1, P2 keep b (StereoMatcher.cpp:402-403)
    //  * a size or cap too large for int16: the largest block that fits with this cap, P1 / P2 following it
    int bs = p.SADWindowSize < 1 ? 1 : p.SADWindowSize, pbs = bs;   // pbs: the size P1 / P2 are computed from
    if ((bs & 1) == 0) {
        bs += 1;
        std::cout << "[StereoMatcherNew] sgbm_hip: even SADWindowSize " << pbs << " -> window " << bs << " (as cv::StereoSGBM), P1/P2 from " << pbs << std::endl;
    }
    int cap = p.preFilterCap < 0 ? 0 : (p.preFilterCap > 127 ? 127 : p.preFilterCap);
    if (cap != p.preFilterCap) std::cout << "[StereoMatcherNew] sgbm_hip: preFilterCap " << p.preFilterCap << " -> " << cap << std::endl;
    const int cn = 1;   // gray cost (x3 with SV_FLAG_COST_RGB, which the module mirror does not set)
    auto fits = [&](int b, int pb) { return b <= 15 && (long)cn * (2 * cap + 63) * b * b + 32L * pb * pb <= 32767; };
    if (!fits(bs, pbs)) {
        const int asked = bs;
        while (bs > 1 && !fits(bs, bs < pbs ? bs : pbs)) bs -= 2;
        if (bs < pbs) pbs = bs;
        std::cout << "[StereoMatcherNew] sgbm_hip: SADWindowSize " << asked << " with preFilterCap " << cap
                  << " would overflow the int16 costs: using " << bs << std::endl;
    }
    hip_params.block_size = bs;
    hip_params.P1 = 8 * pbs * pbs;
    hip_params.P2 = 32 * pbs * pbs;
    hip_params.pre_filter_cap = cap;
    hip_params.uniqueness_ratio = p.uniquenessRatio;
    hip_params.disp12_max_diff = p.disp12MaxDiff;
    // the CPU path filters speckles inside cv::StereoSGBM::compute; the CUDA path has no such stage
    hip_params.speckle_window_size = paths == 8 ? p.speckleWindowSize : 0;
    hip_params.speckle_range = p.speckleRange;
    hip_params.wls_lambda = (float)p.wls_lambda;
    hip_params.wls_sigma = (float)p.wls_sigma;
    if (ctx) {
        last_rc = sv_set_params(ctx, &hip_params);
        if (last_rc) std::cout << "[StereoMatcherNew] !! sgbm_hip rejected the parameters: " << sv_last_error(ctx) << std::endl;
    }
}

bool StereoMatcherNew::ensureContext(int w, int h) {
    if (ctx && (w != ctx_w || h != ctx_h)) { sv_destroy(ctx); ctx = nullptr; mapper_on_device = false; }
    if (!ctx) {
        last_rc = sv_create(&ctx, device, w, h, &hip_params);
        if (last_rc) {
            std::cout << "[StereoMatcherNew] !! sgbm_hip unavailable: " << sv_last_error(ctx) << std::endl;
            if (ctx) sv_destroy(ctx);
            ctx = nullptr;
            return false;
        }
        ctx_w = w; ctx_h = h;
    }
    return true;
}

void StereoMatcherNew::compute() {
    // StereoMatcher.cpp:57-79, unchanged but for the new case
    this->stereo->updateMappings();
    this->stereo->rectifyImages();
    switch (stereo_parameters.stereo_matching) {
        case SGBM_HIP: matchSGBMHIP(); break;
        default:
            std::cout << "[StereoMatcherNew] !! only the sgbm_hip back-end is built into this host shim." << std::endl;
            break;
    }
}

// The new SM_MATCHING_ALG case (README.md:60-118 recipe): ends with "this->disparity16 = result".
// Variant A: like matchSGBMCUDA (StereoMatcher.cpp:420-440) it takes the pair that
// rectifyImages() has rectified on the CPU, through the public getters; StereoCamera is untouched.
// Variant B (StereoCamera patched, setRectifyOnDevice(true)): the raw pair goes up and the two
// cv::initUndistortRectifyMap, the two cv::remap and the cv::undistortPoints of MapperL run on the GPU,
// fed by the matrices cv::stereoRectify left in the camera.
void StereoMatcherNew::matchSGBMHIP() {
    const bool on_device = stereo->getRectifyOnDevice();                                       // PATCH B
    const Image& L = on_device ? stereo->getImLeft() : stereo->getLRectified();
    const Image& R = on_device ? stereo->getImRight() : stereo->getRRectified();
    if (L.empty() || R.empty()) { std::cout << "Images are not set! set the images first!" << std::endl; return; }  // stereoCamera.cpp:554-557
    if (L.channels != 3 || L.elem != 1 || R.rows != L.rows || R.cols != L.cols || R.channels != 3) {
        std::cout << "[StereoMatcherNew] !! image type error" << std::endl;  // bova_sgm.cu:45-49 exits here
        last_rc = SV_ERR_INVALID_ARG;
        return;
    }
    if (!ensureContext(L.cols, L.rows)) return;
    if (on_device && (stereo->takeRectificationChanged() || !mapper_on_device)) {              // PATCH B
        const Image &KL = stereo->getKleft(), &KR = stereo->getKright(), &DL = stereo->getDistCoeffLeft(), &DR = stereo->getDistCoeffRight();
        const Image &RL = stereo->getRLrect(), &RR = stereo->getRRrect(), &PL = stereo->getPLrect(), &PR = stereo->getPRrect();
        last_rc = sv_init_rectify_map(ctx, 0, KL.ptr<double>(), DL.ptr<double>(), (int)DL.total(), RL.ptr<double>(), PL.ptr<double>());
        if (!last_rc) last_rc = sv_init_rectify_map(ctx, 1, KR.ptr<double>(), DR.ptr<double>(), (int)DR.total(), RR.ptr<double>(), PR.ptr<double>());
        if (!last_rc) last_rc = sv_update_mapper(ctx, 0, KL.ptr<double>(), DL.ptr<double>(), (int)DL.total(), RL.ptr<double>(), PL.ptr<double>());
        if (last_rc) { std::cout << "[StereoMatcherNew] !! " << sv_last_error(ctx) << std::endl; return; }
        mapper_on_device = true;
    }
    Image d16(L.rows, L.cols, 1, 2), vis(L.rows, L.cols, 1, 1);
    // the pair, disparity16 and its getDisparityVis(.,.,3) image (StereoMatcher.cpp:437) in one call
    last_rc = sv_compute_rgb_vis_host(ctx, L.ptr<uint8_t>(), L.step(), R.ptr<uint8_t>(), R.step(), on_device ? 1 : 0,
                                      d16.ptr<int16_t>(), d16.step(), vis.ptr<uint8_t>(), vis.step(), 3);
    if (last_rc) { std::cout << "[StereoMatcherNew] !! sgbm_hip failed: " << sv_last_error(ctx) << std::endl; return; }
    this->disparity = vis;
    this->disparity16 = d16;
}

void StereoMatcherNew::filterBLF(std::string kind) {
    Image input = getDisparity(kind);
    Image input16 = getDisparity16(kind);
    switch (stereo_parameters.BLFfiltering) {
        case BLF_ORIGINAL: {
            // StereoMatcher.cpp:90-99: cv_extend::bilateralFilter on the 16-bit map + getDisparityVis(.,.,3)
            if (!ctx || input16.empty()) break;
            Image out16(input16.rows, input16.cols, 1, 2), out8(input16.rows, input16.cols, 1, 1);
            last_rc = sv_filter_blf_host(ctx, input16.ptr<int16_t>(), input16.step(), stereo_parameters.sigmaColorBLF,
                                         stereo_parameters.sigmaSpaceBLF, out16.ptr<int16_t>(), out16.step(), out8.ptr<uint8_t>(),
                                         out8.step());
            if (last_rc != 0) {
                std::cout << "[StereoMatcherNew] !! sv_filter_blf_host failed: " << sv_last_error(ctx) << std::endl;
                break;  // pass the input through, as the disabled branch does
            }
            disparity_blf = out8;
            disparity16_blf = out16;
            return;
        }
        case BLF_CUDA: {
            // StereoMatcher.cpp:101-116: cv::cuda::DisparityBilateralFilter(numberOfDisparities, 7, 2), sigma_range =
            // sigmaColorBLF (:292-301), guided by the gray rectified left image that compute() left on the device
            if (!ctx || input16.empty()) break;
            Image out16(input16.rows, input16.cols, 1, 2), out8(input16.rows, input16.cols, 1, 1);
            last_rc = sv_filter_blf_joint_host(ctx, input16.ptr<int16_t>(), input16.step(), nullptr, 0,
                                               stereo_parameters.numberOfDisparities, 7, 2, stereo_parameters.sigmaColorBLF,
                                               out16.ptr<int16_t>(), out16.step(), out8.ptr<uint8_t>(), out8.step());
            if (last_rc != 0) {
                std::cout << "[StereoMatcherNew] !! sv_filter_blf_joint_host failed: " << sv_last_error(ctx) << std::endl;
                break;
            }
            disparity_blf = out8;
            disparity16_blf = out16;
            return;
        }
        case BLF_DISABLED:
        default:
            break;
    }
    disparity_blf = input.clone();      // StereoMatcher.cpp:117-122
    disparity16_blf = input16.clone();
}

void StereoMatcherNew::filterWLS(std::string kind) {
    Image input = getDisparity(kind);
    Image input16 = getDisparity16(kind);
    switch (stereo_parameters.WLSfiltering) {
        case WLS_ENABLED: {
            // StereoMatcher.cpp:154-160: lambda / sigma / ROI (computeROI2) are taken from the
            // parameters pushed by updateHIPParams(); the guide is the rectified left image, which
            // compute() left on the device (guide pointer NULL).
            if (!ctx || input16.empty() || input.empty()) break;
            Image out16(input16.rows, input16.cols, 1, 2), out8(input.rows, input.cols, 1, 1);
            last_rc = sv_filter_wls_host(ctx, input16.ptr<int16_t>(), input16.step(), input.ptr<uint8_t>(), input.step(), nullptr, 0,
                                         out16.ptr<int16_t>(), out16.step(), out8.ptr<uint8_t>(), out8.step());
            if (last_rc != 0) {
                std::cout << "[StereoMatcherNew] !! sv_filter_wls_host failed: " << sv_last_error(ctx) << std::endl;
                break;  // pass the input through, as the disabled branch does
            }
            disparity_wls = out8;
            disparity16_wls = out16;
            return;
        }
        case WLS_LRCHECK:
            // StereoMatcher.cpp:164-186: only defined for SGBM_OPENCV; for every other matcher the
            // reference clones the input (and, through its missing braces, always clones the 16-bit map)
        case WLS_DISABLED:
        default:
            break;
    }
    disparity_wls = input.clone();      // StereoMatcher.cpp:193-197
    disparity16_wls = input16.clone();
}

Image StereoMatcherNew::getDisparity(std::string kind) {
    if (kind == "base") return disparity;
    if (kind == "blf") return disparity_blf;
    if (kind == "wls") return disparity_wls;
    std::cout << "[StereoMatcherNew] !! Disparity kind " << kind << " not found, returning BASE disparity." << std::endl;
    return disparity;
}

Image StereoMatcherNew::getDisparity16(std::string kind) {
    if (kind == "base") return disparity16;
    if (kind == "blf") return disparity16_blf;
    if (kind == "wls") return disparity16_wls;
    std::cout << "[StereoMatcherNew] !! Disparity16 kind " << kind << " not found, returning BASE disparity16." << std::endl;
    return disparity16;
}

void StereoMatcherNew::setAlgorithm(std::string name) {
    if (name == "sgbm_hip") stereo_parameters.stereo_matching = SGBM_HIP;
    else if (name == "sgbm") stereo_parameters.stereo_matching = SGBM_OPENCV;
    else if (name == "sgbm_cuda") stereo_parameters.stereo_matching = SGBM_CUDA;
    else if (name == "libelas") stereo_parameters.stereo_matching = LIBELAS;
    else {
        std::cout << "[StereoMatcherNew] !! Stereo Matching algorithm " << name << " not found, defaulting to SGBM." << std::endl;
        stereo_parameters.stereo_matching = SGBM_OPENCV;
    }
}

Image StereoMatcherNew::depthFromDisparity(const Image& d16) {
    Image depth(d16.rows, d16.cols, 1, 4);
    if (!ctx || d16.empty()) return depth;
    // variant B: MapperL is on the device already (sv_update_mapper); variant A: the one updateMappings() made
    const Image& mapper = stereo->getMapperL();
    const bool dev_mapper = stereo->getRectifyOnDevice() && mapper_on_device;
    if (!dev_mapper && mapper.empty()) {  // DispModule.cpp:347-348: return disparity/16 unprojected
        for (int r = 0; r < d16.rows; r++)
            for (int c = 0; c < d16.cols; c++) depth.ptr<float>(r)[c] = d16.ptr<int16_t>(r)[c] * (1.f / 16.f);
        return depth;
    }
    last_rc = sv_depth_host(ctx, d16.ptr<int16_t>(), d16.step(), dev_mapper ? nullptr : mapper.ptr<float>(), stereo->getQ().ptr<double>(),
                            stereo->getRLrect().ptr<double>(), depth.ptr<float>(), depth.step());
    if (last_rc) std::cout << "[StereoMatcherNew] !! depth failed: " << sv_last_error(ctx) << std::endl;
    return depth;
}

// ---- StereoCamera test double (see the header) ----
void StereoCamera::setImages(const Image& firstImg, const Image& secondImg) {
    imleft = firstImg;
    imright = secondImg;
}

void StereoCamera::setIntrinsics(const Image& K1, const Image& K2, const Image& Dist1, const Image& Dist2) {
    Kleft = K1; Kright = K2; DistL = Dist1; DistR = Dist2;
    cameraChanged = true;  // stereoCamera.cpp:1949
}

void StereoCamera::setRectification(const Image& RL, const Image& RR, const Image& PL, const Image& PR, const Image& Qm) {
    RLrect = RL; RRrect = RR; PLrect = PL; PRrect = PR; Q = Qm;
    cameraChanged = true;  // the kinematics do this on every frame (DispModule.cpp:684 -> stereoCamera.cpp:1505-1536)
}

void StereoCamera::rectifyImages() {
    if (imleft.empty() || imright.empty()) { std::cout << "Images are not set! set the images first!" << std::endl; return; }  // stereoCamera.cpp:554-557
    // cameraChanged: cv::stereoRectify (stereoCamera.cpp:564) - its result comes from setRectification here
    if (rectifyOnDevice) {                                      // PATCH B: no map generation, no remap on the CPU
        if (cameraChanged) rectificationChanged = true;
        cameraChanged = false;
        return;
    }
    // stereoCamera.cpp:569-574: cv::initUndistortRectifyMap x2 + cv::remap x2.  No OpenCV in this build: the CPU path
    // of this test double serves pairs that are rectified already, for which the remaps are copies.
    imgLeftRect = imleft;
    imgRightRect = imright;
    cameraChanged = false;
}

void StereoCamera::updateMappings() {
    const int rows = imleft.rows, cols = imleft.cols;
    if (!cameraChanged || rectifyOnDevice || rows == 0) return;  // PATCH B: the mapper is made on the device
    // stereoCamera.cpp:2443-2470: cv::undistortPoints on the pixel grid; for a pair that is rectified already the
    // result is the grid itself
    if (MapperL.rows != rows || MapperL.cols != cols) {
        MapperL = Image(rows, cols, 2, 4);
        for (int y = 0; y < rows; y++)
            for (int x = 0; x < cols; x++) { MapperL.ptr<float>(y)[2 * x] = (float)x; MapperL.ptr<float>(y)[2 * x + 1] = (float)y; }
        MapperR = MapperL;
    }
}

}  // namespace svh
