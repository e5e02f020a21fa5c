/*
 * sv_ref.h — CPU oracle (libsv_ref) for the stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (libsv_hip, the host shim); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU comparator.
 *
 * PARITY UNPINNED: the reference (robotology/stereo-vision) ships no tests,
 * golden vectors or known-answer files for this path (SURVEY.md section 4 and
 * 8(c)), OpenCV (which holds the CPU SGBM arithmetic) is not installed, and
 * the sgbm_cuda .cu files need nvcc + OpenCV headers, so it cannot be built here.
 * This file restates
 *   REPO items: the integer arithmetic of modules/DisparityModule/sgbm_cuda
 *               (costs.cu, aggregation.cu, bova_sgm.cu) read as text, and
 *   EXT  items: OpenCV behaviours the reference calls but does not contain
 *               (MODE_HH second sweep, filterSpeckles, remap, getDisparityVis,
 *               cvtColor) restated from their published algorithm.
 * It is pinned only by (a) analytic known-answer tests and (b) an independent
 * second restatement (oracle/naive_sgm.py) — see tests/test_oracle.py.
 */
#ifndef SV_REF_H
#define SV_REF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SVREF_INVALID (-16)

/* same bit values as include/sv_hip.h */
#define SVREF_FLAG_C_COL0_P2 0x01
#define SVREF_FLAG_C_TAIL_P2 0x02
#define SVREF_FLAG_LR_OPENCV 0x04
#define SVREF_FLAG_MEDIAN_REPLICATE 0x08
#define SVREF_FLAG_NO_LRCHECK 0x10
#define SVREF_FLAG_NO_MEDIAN 0x20
#define SVREF_FLAG_COST_RGB 0x40 /* cost summed over R, G, B (cv::StereoSGBM on 8UC3, matchSGBM) */

typedef struct svref_params {
    int32_t min_disparity, num_disparities, block_size, P1, P2, pre_filter_cap;
    int32_t uniqueness_ratio, disp12_max_diff, speckle_window_size, speckle_range;
    uint32_t path_mask, flags;
    float wls_lambda, wls_sigma;
} svref_params;

/* optional stage outputs of svref_sgbm (any pointer may be NULL) */
typedef struct svref_stages {
    uint8_t* grad_l;   /* [H][W]        A1  */
    uint8_t* grad_r;   /* [H][W]            */
    uint8_t* raw_l;    /* [H][W] raw channel with edge columns = ftzero */
    uint8_t* raw_r;
    int16_t* pix;      /* [H][W1][D]    A2  */
    int16_t* hsum;     /* [H][W1][D]    A3  */
    int16_t* C;        /* [H][W1][D]    A4  */
    int16_t* L[8];     /* [H][W1][D] per path, A5 */
    int16_t* S;        /* [H][W1][D]        */
    int16_t* disp_wta; /* [H][W]        A6 (before LR) */
    int16_t* raw;      /* [H][W]  integer best d, -1 invalid */
    int16_t* mins;     /* [H][W]            */
    int16_t* disp2;    /* [H][W]        A7  */
    int16_t* disp_lr;  /* [H][W]  after LR check, before median */
    int16_t* disp_med; /* [H][W]  after median, before speckle */
} svref_stages;

int svref_check_params(const svref_params* p, int W, int H);

/* A1: costs.cu:116-166, :286-292 */
void svref_prefilter(const uint8_t* img, int W, int H, int cap, uint8_t* grad, uint8_t* raw_mod);
/* A2-A4: costs.cu:170-284 */
int svref_cost_volume(const uint8_t* left, const uint8_t* right, int W, int H, const svref_params* p,
                      int16_t* C, svref_stages* st);
/* A5: aggregation.cu:60-625.  dir in 0..7 (bit index of SV_PATH_*).  S is
 * accumulated with saturation; L_out may be NULL.                           */
void svref_aggregate_path(const int16_t* C, int W1, int H, int D, int P1, int P2, int dir, int16_t* S,
                          int16_t* L_out);
/* A6: aggregation.cu:629-689 */
void svref_wta(const int16_t* S, int W, int H, int D, int uniq, int16_t* disp, int16_t* raw, int16_t* mins);
/* A7: aggregation.cu:691-735 */
void svref_lrcheck(int16_t* disp, const int16_t* raw, const int16_t* mins, int W, int H, int D, int thresh,
                   int16_t* disp2_out);
/* A6 / A7 with min_disparity (EXT, cv::StereoSGBM as matchSGBM parameterises it, StereoMatcher.cpp:399): the
 * computable columns are [max(minD + D, 0), W + min(minD, 0)), cost index d pairs left x with right
 * x - d - minD, stored disparity = (d + minD) * 16 + sub-pixel, invalid = (minD - 1) * 16; raw stays the index */
void svref_wta_min(const int16_t* S, int W, int H, int D, int minD, int uniq, int16_t* disp, int16_t* raw, int16_t* mins);
void svref_lrcheck_min(int16_t* disp, const int16_t* raw, const int16_t* mins, int W, int H, int D, int minD, int thresh,
                       int16_t* disp2_out);
/* A8: aggregation.cu:737-768 */
void svref_median3(const int16_t* in, int16_t* out, int W, int H, int replicate);
/* A10 (EXT cv::filterSpeckles) */
void svref_filter_speckles(int16_t* img, int W, int H, int new_val, int max_size, int max_diff);
/* whole matcher, gray 8-bit pair -> disparity16; strides in bytes */
int svref_sgbm(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int W, int H,
               const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st);

/* matchSGBM (StereoMatcher.cpp:392-417), EXT: the same matcher on interleaved 8UC3 images with the BT
 * cost of calcPixelCostBT summed over the three colour planes; needs SVREF_FLAG_COST_RGB */
int svref_sgbm_rgb(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int W, int H,
                   const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st);

/* A11 (EXT getDisparityVis) */
void svref_disparity_vis(const int16_t* disp16, uint8_t* vis, size_t n, int scale);
/* A12 (EXT cvtColor BGR2GRAY on RGB-ordered bytes, StereoMatcher.cpp:429) */
void svref_gray_from_rgb(const uint8_t* rgb, uint8_t* gray, size_t n);
/* A13 (EXT cv::remap INTER_LINEAR, BORDER_CONSTANT 0), 8UC3; stereoCamera.cpp:573 */
void svref_remap_rgb(const uint8_t* src, int W, int H, const float* mapx, const float* mapy, uint8_t* dst);
/* A14 DispModule.cpp:320-366.  alias = 1 reproduces the in-place read of the
 * reference; alias = 0 reads the pristine disparity (Appendix B-7).          */
void svref_depth(const int16_t* disp16, int W, int H, const float* mapper, const double* Q, const double* R,
                 int alias, float* depth);

/* A15 (EXT createDisparityWLSFilterGeneric(false) -> fastGlobalSmootherFilter), StereoMatcher.cpp:127-198 */
void svref_wls_lut(float sigma, float* lut /* [3*256*256] */);
void svref_wls_roi(int W, int H, const svref_params* p, int roi[4]); /* computeROI2, StereoMatcher.cpp:12-29 */
int svref_fgs(const uint8_t* guide, size_t gs, float* img, int w, int h, float lambda, float sigma, float att,
              int iters);
int svref_wls_filter(const int16_t* disp16, const uint8_t* vis, const uint8_t* guide_rgb, int W, int H, const int roi[4],
                     float lambda, float sigma, int16_t* out16, uint8_t* out8);

/* EXT cv::initUndistortRectifyMap (stereoCamera.cpp:569-570) and cv::undistortPoints on the pixel grid
 * (stereoCamera.cpp:2462-2463).  K 3x3, dist {k1,k2,p1,p2,k3,k4,k5,k6}, R 3x3, P 3x4, row-major doubles. */
int svref_init_rectify_map(const double* K, const double* dist, const double* R, const double* P, int W, int H,
                           float* mapx, float* mapy);
int svref_undistort_grid(const double* K, const double* dist, const double* R, const double* P, int W, int H,
                         float* mapper);

/* A16 filterBLF / BLF_ORIGINAL: cv_extend::bilateralFilter (fastBilateral.hpp:92-189, in the reference tree) */
int svref_blf(const int16_t* src, int W, int H, double sigma_color, double sigma_space, int16_t* dst);

/* 8(f)2 world image: SFM.cpp:1284-1347 with the point arithmetic of DispModule.cpp:1103-1160 */
void svref_world_image(const int16_t* disp16, int W, int H, const float* mapper, const double* Q, const double* RLrect,
                       const double* HL_root, float* cart, float* cyl);

/* A16 filterBLF / BLF_CUDA (EXT): cv::cuda::DisparityBilateralFilter with a gray guide, deterministic schedule */
int svref_blf_joint(const int16_t* disp, const uint8_t* guide, int W, int H, int ndisp, int radius, int iters, double sigma_range,
                    int16_t* out);

const char* svref_version(void);
#ifdef __cplusplus
}
#endif
#endif
