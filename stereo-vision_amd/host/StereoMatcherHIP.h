// StereoMatcherHIP.h — host-side C++ mirror of the reference's matcher class for the SGBM_HIP
// back-end.  Same names, argument meaning and error behaviour as
// /root/reference/modules/DisparityModule/StereoMatcher.h:27-238, minus everything the hot path
// does not touch (LIBELAS, OpenCV SGBM, CUDA BLF), with cv::Mat replaced by a small owned image
// type because OpenCV/YARP are not available in this build environment.  In the real module the
// only addition is matchSGBMHIP() + the enum value; this file shows that code
// compiling and running against libsv_hip.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sv_hip.h"

namespace svh {

// Stand-in for a continuous cv::Mat: CV_8UC1 / CV_8UC3 / CV_16SC1 / CV_32FC1 / CV_32FC2.
struct Image {
    int rows = 0, cols = 0, channels = 1, elem = 1;  // elem = bytes per channel element
    std::vector<uint8_t> data;
    Image() = default;
    Image(int r, int c, int ch, int e) : rows(r), cols(c), channels(ch), elem(e), data((size_t)r * c * ch * e) {}
    bool empty() const { return data.empty(); }
    size_t step() const { return (size_t)cols * channels * elem; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data.data() + (size_t)r * step()); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data.data() + (size_t)r * step()); }
    Image clone() const { return *this; }
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

// The slice of StereoCamera (lib/include/iCub/stereoVision/stereoCamera.h) the matcher reads:
// images, rectification maps (what initUndistortRectifyMap produced, stereoCamera.cpp:569-570), Q,
// RLrect, MapperL.  Map generation itself stays with the caller.
class StereoCamera {
public:
    void setImages(const Image& left, const Image& right) { imleft = left; imright = right; }
    void setRectifyMaps(const Image& m11, const Image& m12, const Image& m21, const Image& m22) {
        map11 = m11; map12 = m12; map21 = m21; map22 = m22; cameraChanged = true;
    }
    void setQ(const double q[16]) { for (int i = 0; i < 16; i++) Q[i] = q[i]; }
    void setRLrect(const double r[9]) { for (int i = 0; i < 9; i++) RLrect[i] = r[i]; }
    void setMapperL(const Image& m) { MapperL = m; }
    const Image& getImLeft() const { return imleft; }
    const Image& getImRight() const { return imright; }
    const Image& getLRectified() const { return imgLeftRect; }
    const Image& getRRectified() const { return imgRightRect; }
    const double* getQ() const { return Q; }
    const double* getRLrect() const { return RLrect; }
    const Image& getMapperL() const { return MapperL; }
    bool hasMaps() const { return !map11.empty(); }

    Image imleft, imright, imgLeftRect, imgRightRect, map11, map12, map21, map22, MapperL;
    double Q[16] = {0}, RLrect[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool cameraChanged = false;  // set by setRotation/setTranslation in the reference (stereoCamera.cpp:1505-1536)
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
    bool maps_uploaded = false;
    int last_rc = 0;
    Image disparity, disparity16, disparity_blf, disparity16_blf, disparity_wls, disparity16_wls;
    Params stereo_parameters;
    sv_params hip_params{};
};

}  // namespace svh
