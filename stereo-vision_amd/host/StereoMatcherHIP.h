// StereoMatcherHIP.h — host-side C++ mirror of the reference's matcher class for the SGBM_HIP
// back-end.  Same names, argument meaning and error behaviour as
// /root/reference/modules/DisparityModule/StereoMatcher.h:27-238, minus everything the hot path
// does not touch (LIBELAS, OpenCV SGBM, CUDA BLF), with cv::Mat replaced by a small owned image
// type because OpenCV/YARP are not available in this build environment.  In the real module the
// only addition is matchSGBMHIP() + the enum value; this file shows that code
// compiling and running against libsv_hip.
//
// The StereoCamera below exposes ONLY what the reference's class makes public
// (lib/include/iCub/stereoVision/stereoCamera.h:181-600; each member cites its line), so that
// matchSGBMHIP() as written here compiles against the unmodified reference (INTEGRATION.md,
// variant A).  The four members marked PATCH B are the additions of INTEGRATION.md's variant B
// (maps, remap and mapper on the GPU) and exist in the reference only after that patch.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sv_hip.h"

namespace svh {

// Stand-in for a continuous cv::Mat: CV_8UC1 / CV_8UC3 / CV_16SC1 / CV_32FC1 / CV_32FC2 / CV_64FC1.
struct Image {
    int rows = 0, cols = 0, channels = 1, elem = 1;  // elem = bytes per channel element
    std::vector<uint8_t> data;
    Image() = default;
    Image(int r, int c, int ch, int e) : rows(r), cols(c), channels(ch), elem(e), data((size_t)r * c * ch * e) {}
    bool empty() const { return data.empty(); }
    size_t step() const { return (size_t)cols * channels * elem; }
    size_t total() const { return (size_t)rows * cols; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data.data() + (size_t)r * step()); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data.data() + (size_t)r * step()); }
    Image clone() const { return *this; }
    static Image mat64(int r, int c, const double* v) {
        Image m(r, c, 1, 8);
        for (int i = 0; i < r * c; i++) m.ptr<double>()[i] = v[i];
        return m;
    }
};

// StereoMatcher.h:27-49 with the new back-end appended so persisted values 0/1/2 keep their meaning
// (DispModule.cpp:915,1068 store the integer in SFM.ini).
enum SM_MATCHING_ALG { SGBM_OPENCV = 0, SGBM_CUDA, LIBELAS, SGBM_HIP };
enum SM_BLF_FILTER { BLF_DISABLED = 0, BLF_ORIGINAL, BLF_CUDA };
enum SM_WLS_FILTER { WLS_DISABLED = 0, WLS_ENABLED, WLS_LRCHECK };

// StereoMatcher.h:55-75
struct Params {
    int uniquenessRatio = 15, speckleWindowSize = 50, speckleRange = 16, numberOfDisparities = 96;
    int SADWindowSize = 7, minDisparity = 0, preFilterCap = 63, disp12MaxDiff = 0;
    double sigmaColorBLF = 10.0, sigmaSpaceBLF = 10.0, wls_lambda = 4500., wls_sigma = 1.0;
    SM_BLF_FILTER BLFfiltering = BLF_DISABLED;
    SM_WLS_FILTER WLSfiltering = WLS_DISABLED;
    SM_MATCHING_ALG stereo_matching = SGBM_HIP;
};

// The PUBLIC slice of StereoCamera the matcher path uses; data members are private as in the
// reference (stereoCamera.h:93-178).  Test double: without OpenCV this class cannot run
// cv::stereoRectify / cv::initUndistortRectifyMap / cv::remap / cv::undistortPoints, so
//  * the result of cv::stereoRectify (stereoCamera.cpp:564) is handed in by the harness
//    (setRectification, not a member of the reference), and
//  * on the CPU path (variant A) rectifyImages() / updateMappings() only serve pairs that are
//    rectified already: the two remaps are copies and the mapper is the identity grid.
class StereoCamera {
public:
    void setImages(const Image& firstImg, const Image& secondImg);                   // stereoCamera.h:238
    void setIntrinsics(const Image& K1, const Image& K2, const Image& Dist1, const Image& Dist2);  // :501
    void rectifyImages();                                                            // :507  (stereoCamera.cpp:548-578)
    void updateMappings();                                                           // stereoCamera.cpp:2437-2472
    const Image& getImLeft() const { return imleft; }                                // :370
    const Image& getImRight() const { return imright; }                              // :376
    const Image& getLRectified() const { return imgLeftRect; }                       // :513
    const Image& getRRectified() const { return imgRightRect; }                      // :519
    const Image& getQ() const { return Q; }                                          // :406
    const Image& getKleft() const { return Kleft; }                                  // :412
    const Image& getKright() const { return Kright; }                                // :418
    const Image& getDistCoeffLeft() const { return DistL; }                          // :547
    const Image& getDistCoeffRight() const { return DistR; }                         // :541
    const Image& getRLrect() const { return RLrect; }                                // :466
    const Image& getRRrect() const { return RRrect; }                                // :472
    const Image& getMapperL() const { return MapperL; }                              // :454
    const Image& getMapperR() const { return MapperR; }                              // :460

    // ---- PATCH B (variant B): not in the reference ----
    const Image& getPLrect() const { return PLrect; }
    const Image& getPRrect() const { return PRrect; }
    void setRectifyOnDevice(bool on) { rectifyOnDevice = on; }  // rectifyImages() keeps cv::stereoRectify, skips map generation and remap
    bool takeRectificationChanged() { const bool c = rectificationChanged; rectificationChanged = false; return c; }
    bool getRectifyOnDevice() const { return rectifyOnDevice; }
    void setMapperL(const Image& m) { MapperL = m; }  // what sv_get_mapper_host read back (getMapperL users: get3DPoints, DispModule.cpp:1103-1197)

    // ---- test double only: what cv::stereoRectify writes at stereoCamera.cpp:564, and the cameraChanged flag that
    // setRotation / setTranslation raise on every frame (stereoCamera.cpp:1505-1536) ----
    void setRectification(const Image& RL, const Image& RR, const Image& PL, const Image& PR, const Image& Qm);

private:
    Image imleft, imright, imgLeftRect, imgRightRect;
    Image Kleft, Kright, DistL, DistR, RLrect, RRrect, PLrect, PRrect, Q, MapperL, MapperR;
    bool cameraChanged = true;
    bool rectifyOnDevice = false, rectificationChanged = false;  // PATCH B
};

class StereoMatcherNew {
public:
    StereoMatcherNew(StereoCamera* stereo, int device = 0);
    ~StereoMatcherNew();
    StereoMatcherNew(const StereoMatcherNew&) = delete;
    StereoMatcherNew& operator=(const StereoMatcherNew&) = delete;

    // StereoMatcher.cpp:32-54, same 15 arguments in the same order
    void setParameters(int minDisparity, int numberOfDisparities, int SADWindowSize, int disp12MaxDiff, int preFilterCap,
                       int uniquenessRatio, int speckleWindowSize, int speckleRange, double sigmaColorBLF,
                       double sigmaSpaceBLF, double wls_lambda, double wls_sigma, SM_BLF_FILTER BLFfiltering,
                       SM_WLS_FILTER WLSfiltering, SM_MATCHING_ALG stereo_matching);
    void compute();                          // StereoMatcher.cpp:57-79
    void setAlgorithm(std::string name);     // StereoMatcher.cpp:240-256 (+ "sgbm_hip")
    Image getDisparity(std::string kind = "base");    // StereoMatcher.cpp:201-217
    Image getDisparity16(std::string kind = "base");  // StereoMatcher.cpp:221-237
    void updateHIPParams();                  // counterpart of updateCUDAParams, StereoMatcher.cpp:271-289
    void filterBLF(std::string kind = "base");  // StereoMatcher.cpp:82-124 (disabled -> clone)
    void filterWLS(std::string kind = "base");  // StereoMatcher.cpp:127-198 (disabled -> clone)
    // DispModule::depthFromDisparity (DispModule.cpp:320-366) on the GPU
    Image depthFromDisparity(const Image& disparity16);
    bool ok() const { return last_rc == 0; }
    int paths = 8;  // 5 = the CUDA back-end's variant, 8 = MODE_HH (DispModule::useBestDisp, DispModule.cpp:371)

private:
    void matchSGBMHIP();  // the new SM_MATCHING_ALG case
    bool ensureContext(int width, int height);
    StereoCamera* stereo;
    int device;
    sv_ctx* ctx = nullptr;
    int ctx_w = 0, ctx_h = 0;
    bool mapper_on_device = false;
    int last_rc = 0;
    Image disparity, disparity16, disparity_blf, disparity16_blf, disparity_wls, disparity16_wls;
    Params stereo_parameters;
    sv_params hip_params{};
};

}  // namespace svh
