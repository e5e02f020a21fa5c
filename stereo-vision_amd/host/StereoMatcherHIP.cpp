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
    hip_params.block_size = p.SADWindowSize;
    hip_params.P1 = 8 * p.SADWindowSize * p.SADWindowSize;
    hip_params.P2 = 32 * p.SADWindowSize * p.SADWindowSize;
    hip_params.pre_filter_cap = p.preFilterCap;
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
    if (ctx && (w != ctx_w || h != ctx_h)) { sv_destroy(ctx); ctx = nullptr; maps_uploaded = false; }
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
    // updateMappings()/rectifyImages() (StereoMatcher.cpp:62-63): the maps come from the caller;
    // the two cv::remap calls (stereoCamera.cpp:573-574) run on the GPU inside matchSGBMHIP.
    switch (stereo_parameters.stereo_matching) {
        case SGBM_HIP: matchSGBMHIP(); break;
        default:
            std::cout << "[StereoMatcherNew] !! only the sgbm_hip back-end is built into this host shim." << std::endl;
            break;
    }
}

// The new SM_MATCHING_ALG case (README.md:60-118 recipe): ends with "this->disparity16 = result".
void StereoMatcherNew::matchSGBMHIP() {
    const Image& L = stereo->getImLeft();
    const Image& R = stereo->getImRight();
    if (L.empty() || R.empty()) { std::cout << "Images are not set! set the images first!" << std::endl; return; }  // stereoCamera.cpp:554-557
    if (L.channels != 3 || L.elem != 1 || R.rows != L.rows || R.cols != L.cols || R.channels != 3) {
        std::cout << "[StereoMatcherNew] !! image type error" << std::endl;  // bova_sgm.cu:45-49 exits here
        last_rc = SV_ERR_INVALID_ARG;
        return;
    }
    if (!ensureContext(L.cols, L.rows)) return;
    const bool rectify = stereo->hasMaps();
    if (rectify && (stereo->cameraChanged || !maps_uploaded)) {
        last_rc = sv_upload_maps_host(ctx, stereo->map11.ptr<float>(), stereo->map12.ptr<float>(), stereo->map21.ptr<float>(),
                                      stereo->map22.ptr<float>());
        if (last_rc) { std::cout << "[StereoMatcherNew] !! " << sv_last_error(ctx) << std::endl; return; }
        maps_uploaded = true;
        stereo->cameraChanged = false;  // stereoCamera.cpp:576
    }
    Image d16(L.rows, L.cols, 1, 2);
    last_rc = sv_compute_rgb_host(ctx, L.ptr<uint8_t>(), L.step(), R.ptr<uint8_t>(), R.step(), rectify ? 1 : 0,
                                  d16.ptr<int16_t>(), d16.step());
    if (last_rc) { std::cout << "[StereoMatcherNew] !! sgbm_hip failed: " << sv_last_error(ctx) << std::endl; return; }
    stereo->imgLeftRect = Image(L.rows, L.cols, 3, 1);
    stereo->imgRightRect = Image(L.rows, L.cols, 3, 1);
    sv_get_rectified_host(ctx, 0, stereo->imgLeftRect.ptr<uint8_t>(), stereo->imgLeftRect.step());
    sv_get_rectified_host(ctx, 1, stereo->imgRightRect.ptr<uint8_t>(), stereo->imgRightRect.step());
    Image vis(L.rows, L.cols, 1, 1);
    last_rc = sv_disparity_vis_host(ctx, d16.ptr<int16_t>(), vis.ptr<uint8_t>(), 3);  // getDisparityVis(.,.,3), StereoMatcher.cpp:437
    if (last_rc) { std::cout << "[StereoMatcherNew] !! " << sv_last_error(ctx) << std::endl; return; }
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
    const Image& mapper = stereo->getMapperL();
    if (!ctx || d16.empty()) return depth;
    if (mapper.empty()) {  // DispModule.cpp:347-348: return disparity/16 unprojected
        for (int r = 0; r < d16.rows; r++)
            for (int c = 0; c < d16.cols; c++) depth.ptr<float>(r)[c] = d16.ptr<int16_t>(r)[c] * (1.f / 16.f);
        return depth;
    }
    last_rc = sv_depth_host(ctx, d16.ptr<int16_t>(), d16.step(), mapper.ptr<float>(), stereo->getQ(), stereo->getRLrect(),
                            depth.ptr<float>(), depth.step());
    if (last_rc) std::cout << "[StereoMatcherNew] !! depth failed: " << sv_last_error(ctx) << std::endl;
    return depth;
}

}  // namespace svh
