/*
 * sv_hip.h — C ABI of libsv_hip, the MI355X (gfx950) stereo disparity backend.
 *
 * This is the drop-in boundary for ONE path of robotology/stereo-vision:
 *   StereoMatcherNew::compute() -> rectify -> SGM disparity16 -> LR / median /
 *   speckle -> depth   (reference: modules/DisparityModule/StereoMatcher.cpp:57-79).
 *
 * The reference's own GPU back-end is bound through three plain functions
 *   void    cuda_init(SGM_PARAMS*)                    sgbm_cuda/bova_sgm.cu:33-41
 *   cv::Mat compute_disparity(cv::Mat*,cv::Mat*,float*) sgbm_cuda/bova_sgm.cu:43-328
 *   cv::Mat zy_remap(cv::Mat&,cv::Mat&)               sgbm_cuda/main.cpp:106-109
 * declared in sgbm_cuda_header.h and called from StereoMatcherNew::matchSGBMCUDA()
 * (StereoMatcher.cpp:420-440) and updateCUDAParams() (StereoMatcher.cpp:271-289).
 * libsv_hip replaces exactly those three entry points (plus the OpenCV calls
 * that surround them on the path) with an opaque-context C ABI: plain pointers
 * and sizes, no cv::Mat, no torch types, no global state, no exit().
 *
 * Conventions
 *   - every call returns 0 on success, a negative sv_status on failure;
 *     sv_last_error(ctx) returns a human-readable string (never NULL).
 *   - pointer arguments named *_dev are DEVICE pointers (HBM) on the context's
 *     GPU; *_host are host pointers.  Strides are in BYTES.
 *   - a context is single-threaded (the reference serialises compute() under
 *     mutexDisp, DispModule.cpp:713-717); different contexts may be used from
 *     different threads / streams concurrently.
 *   - all work of a context is enqueued on the context's HIP stream
 *     (sv_set_stream); *_host entry points synchronise that stream before
 *     returning, *_dev entry points do not.
 */
#ifndef SV_HIP_H
#define SV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SV_ABI_VERSION 1

/* disparity16 contract: StereoMatcher.h:105-106, sgbm_cuda/common.h:17-19 */
#define SV_DISP_SCALE 16
#define SV_DISP_SHIFT 4
#define SV_INVALID_DISP_SCALED (-16)

typedef enum sv_status {
    SV_OK = 0,
    SV_ERR_INVALID_ARG = -1,  /* NULL pointer, bad size, unsupported parameter  */
    SV_ERR_UNSUPPORTED = -2,  /* parameter combination the kernels do not cover */
    SV_ERR_NO_DEVICE = -3,    /* no HIP device / wrong architecture             */
    SV_ERR_HIP = -4,          /* a HIP runtime call failed (see sv_last_error)  */
    SV_ERR_NO_MAPS = -5,      /* rectify / depth requested before maps uploaded */
    SV_ERR_NOMEM = -6
} sv_status;

/* Direction bits of the SGM paths (predecessor of pixel p is p - r).           */
/* SURVEY Appendix A5; reference kernels aggregation.cu:60 (LR) :147 (UD_LR)    */
/* :308 (UD) :388 (RL_UD) :546 (RL); the last three bits are the second sweep   */
/* of OpenCV MODE_HH (StereoMatcher.cpp:409).                                   */
#define SV_PATH_R 0x01  /* r=(+1, 0)  left  -> right                           */
#define SV_PATH_DR 0x02 /* r=(+1,+1)  down-right                               */
#define SV_PATH_D 0x04  /* r=( 0,+1)  top   -> bottom                          */
#define SV_PATH_DL 0x08 /* r=(-1,+1)  down-left                                */
#define SV_PATH_L 0x10  /* r=(-1, 0)  right -> left                            */
#define SV_PATH_UL 0x20 /* r=(-1,-1)  up-left                                  */
#define SV_PATH_U 0x40  /* r=( 0,-1)  bottom-> top                             */
#define SV_PATH_UR 0x80 /* r=(+1,-1)  up-right                                 */
#define SV_PATHS_5 0x1F /* the reference CUDA back-end (bova_sgm.cu:108-126)   */
#define SV_PATHS_8 0xFF /* MODE_HH, what matchSGBM() asks for                  */

/* Variant flags: each selects between the in-repo   */
/* CUDA statement (REPO) and the restated OpenCV behaviour (EXT).               */
#define SV_FLAG_C_COL0_P2 0x01      /* C[y>0][x=D] = P2 (MODE_HH) instead of a copy of row 0 (costs.cu:254-267) */
#define SV_FLAG_C_TAIL_P2 0x02      /* C[y>H-1-SH2] = P2 (MODE_HH) instead of a copy of the previous row (costs.cu:279-282) */
#define SV_FLAG_LR_OPENCV 0x04      /* LR threshold = disp12MaxDiff>0 ? disp12MaxDiff : 1 instead of the raw value (aggregation.cu:732) */
#define SV_FLAG_MEDIAN_REPLICATE 0x08 /* cv::medianBlur border replication instead of unfiltered borders (aggregation.cu:765-767) */
#define SV_FLAG_NO_LRCHECK 0x10     /* skip the left-right check (disp12MaxDiff < 0 in OpenCV) */
#define SV_FLAG_NO_MEDIAN 0x20      /* skip the 3x3 median */
#define SV_FLAG_COST_RGB 0x40       /* matching cost summed over R, G, B: cv::StereoSGBM on the 8UC3 pair, as matchSGBM runs it
                                      (StereoMatcher.cpp:392-417; P1 = 8*3*SAD^2, P2 = 32*3*SAD^2 are the caller's to set).  Only the
                                      sv_compute_rgb_* entry points accept it; block sizes whose colour cost would wrap int16
                                      (> 7 at cap 63) are rejected */
#define SV_FLAGS_REPO 0x00          /* SGBM5_REPO: bit-faithful to the sgbm_cuda kernels */
#define SV_FLAGS_HH (SV_FLAG_C_COL0_P2 | SV_FLAG_C_TAIL_P2 | SV_FLAG_LR_OPENCV | SV_FLAG_MEDIAN_REPLICATE)

/*
 * Parameter block.  Replaces SGM_PARAMS (sgbm_cuda/common.h:27-36) and the
 * matcher part of Params (StereoMatcher.h:55-75).  Unlike the reference,
 * num_disparities is honoured at run time (the CUDA code ignores it,
 * common.h:16) and pre_filter_cap changes take effect (bova_sgm.cu:51-78
 * builds the clip table only once).
 */
typedef struct sv_params {
    int32_t min_disparity;       /* minDisparity of cv::StereoSGBM (StereoMatcher.cpp:399; RPC setMinDisp, DispModule.cpp:1355-1420;
                                    module default 0, DispModule.cpp:377), -1024..1024.  EXT item (the CUDA back-end has no such
                                    parameter): computable columns [max(minD + D, 0), W + min(minD, 0)), cost index d pairs left
                                    x with right x - d - minD, disparity16 = (d + minD) * 16 + sub-pixel, every other pixel is
                                    (minD - 1) * 16 (= SV_INVALID_DISP_SCALED for minD = 0)                                  */
    int32_t num_disparities;     /* D: multiple of 32, 32..256 (DispModule.cpp:1355-1420) */
    int32_t block_size;          /* SADWindowSize, odd, 1..15 (subject to the int16 bound: sv_set_params) */
    int32_t P1;                  /* 8*BS*BS  (StereoMatcher.cpp:279)                     */
    int32_t P2;                  /* 32*BS*BS (StereoMatcher.cpp:280)                     */
    int32_t pre_filter_cap;      /* 0..127 (2 * cap fits the reference's u8 clip table)  */
    int32_t uniqueness_ratio;    /* 0..100                                               */
    int32_t disp12_max_diff;     /* LR threshold, see SV_FLAG_LR_OPENCV                  */
    int32_t speckle_window_size; /* 0 = off (reference CUDA path has none)               */
    int32_t speckle_range;       /* maxDiff = 16*speckle_range                           */
    uint32_t path_mask;          /* SV_PATHS_5 / SV_PATHS_8 / any non-empty subset       */
    uint32_t flags;              /* SV_FLAG_*                                            */
    float wls_lambda;            /* StereoMatcher.cpp:154                                */
    float wls_sigma;             /* StereoMatcher.cpp:155                                */
} sv_params;

typedef struct sv_ctx sv_ctx;

/* Fill *p with the module defaults (DispModule.cpp:369-394, numDisparities as
 * given) for the gray/x1 cost variant of the CUDA back-end.  paths = 5 or 8. */
int sv_default_params(sv_params* p, int num_disparities, int paths);

/* Replaces the lazy first-call allocation of compute_disparity
 * (bova_sgm.cu:51-78).  Allocates every device buffer for a width x height
 * frame on HIP device `device`.  The context owns them until sv_destroy. */
int sv_create(sv_ctx** out, int device, int width, int height, const sv_params* params);

/* Replaces free_gpu_mem (bova_sgm.cu:331-351, never called by the reference). */
int sv_destroy(sv_ctx* ctx);

/* Replaces cuda_init(SGM_PARAMS*) (bova_sgm.cu:33-41) as called from
 * updateCUDAParams (StereoMatcher.cpp:271-289).  Reallocates if D changed.   */
int sv_set_params(sv_ctx* ctx, const sv_params* params);
int sv_get_params(const sv_ctx* ctx, sv_params* params);

/* All later work of ctx is enqueued on `hip_stream` (a hipStream_t; NULL = the
 * default stream).  No counterpart in the reference (it uses the default
 * stream and cudaDeviceSynchronize after every kernel, bova_sgm.cu:92-130).  */
int sv_set_stream(sv_ctx* ctx, void* hip_stream);

/* Waits for everything enqueued on the context stream and reports deferred
 * device-side errors of the *_dev entry points (the reference synchronises
 * after every kernel instead, bova_sgm.cu:92-130).                            */
int sv_synchronize(sv_ctx* ctx);

/* Replaces zy_remap / compute_disparity (main.cpp:106-109, bova_sgm.cu:43-328):
 * 8-bit gray rectified pair in host memory -> disparity16 in host memory.
 * Columns [0, D) (min_disparity = 0; see sv_params for other values) are written as
 * SV_INVALID_DISP_SCALED (the reference leaves them uninitialised, bova_sgm.cu:67). */
int sv_compute_gray_host(sv_ctx* ctx, const uint8_t* left_host, size_t left_stride,
                         const uint8_t* right_host, size_t right_stride,
                         int16_t* disp16_host, size_t disp_stride);

/* The same, asynchronously: sv_submit_gray_host stages the pair in pinned memory,
 * enqueues upload + matcher + download on the context stream and returns a
 * ticket at once; sv_wait_host blocks until that pair is done and copies its
 * disparity16 out.  Up to SV_MAX_IN_FLIGHT pairs may be in flight, so the
 * copies and the host side of frame i+1 overlap the kernels of frame i (the
 * reference blocks on every copy and kernel, bova_sgm.cu:87-136).  Tickets
 * must be waited for in the order they were issued.                           */
#define SV_MAX_IN_FLIGHT 8
int sv_submit_gray_host(sv_ctx* ctx, const uint8_t* left_host, size_t left_stride,
                        const uint8_t* right_host, size_t right_stride, int* ticket);
int sv_wait_host(sv_ctx* ctx, int ticket, int16_t* disp16_host, size_t disp_stride);

/* Many pairs per call through host buffers, for hosts that serve several camera
 * streams (BASELINE.json configs[4]; DispModule itself handles one pair per
 * updateModule(), DispModule.cpp:661-814).  sv_submit_gray_batch_host enqueues,
 * for n_frames <= sv_set_batch pairs given as arrays of per-frame pointers
 * (frames of different streams live in different buffers): the uploads on a copy
 * stream of the context, one batched matcher launch on the context stream, and
 * the downloads into disp16_host[i] on a second copy stream; it returns a ticket
 * at once.  SV_BATCH_SLOTS batches may be in flight, so the PCIe traffic of batch
 * k+1 and k-1 overlaps the kernels of batch k.  Nothing is staged: the copies
 * run at PCIe speed and asynchronously when the buffers are pinned
 * (sv_host_alloc, or hipHostMalloc / hipHostRegister by the caller) and through
 * the runtime's own staging otherwise.  All buffers, inputs and outputs, must
 * stay valid and untouched until sv_wait_batch_host(ticket) returns; tickets are
 * waited for in the order they were issued.  No counterpart in the reference,
 * which blocks on every copy (bova_sgm.cu:87-136).                            */
#define SV_BATCH_SLOTS 2
int sv_submit_gray_batch_host(sv_ctx* ctx, int n_frames, const uint8_t* const* left_host, size_t left_stride,
                              const uint8_t* const* right_host, size_t right_stride,
                              int16_t* const* disp16_host, size_t disp_stride, int* ticket);
int sv_wait_batch_host(sv_ctx* ctx, int ticket);
/* Pinned (page-locked) host memory for the buffers above, for hosts that do not
 * link the HIP runtime themselves.                                            */
int sv_host_alloc(void** ptr, size_t bytes);
int sv_host_free(void* ptr);

/* Same path with inputs and output already resident in HBM (dense rows:
 * stride = width resp. 2*width bytes).  Asynchronous on the context stream.   */
int sv_compute_gray_dev(sv_ctx* ctx, const uint8_t* left_dev, const uint8_t* right_dev,
                        int16_t* disp16_dev);

/* Frame batching.  Stereo pairs are independent, so one launch can
 * carry many of them: sv_set_batch sizes the internal volumes for up to
 * max_frames pairs, 1 .. 2048 (288 GB of HBM hold 2000 640x480 D=128 frames) and
 * sv_compute_gray_batch_dev runs n_frames dense-packed pairs ([n][H][W] u8 in,
 * [n][H][W] int16 out) through the same kernels with the frame index as a grid
 * dimension.  No counterpart in the reference (one frame, static buffers,
 * bova_sgm.cu:7-30); the single-frame entry points are the batch of one.      */
int sv_set_batch(sv_ctx* ctx, int max_frames);
int sv_compute_gray_batch_dev(sv_ctx* ctx, int n_frames, const uint8_t* left_dev, const uint8_t* right_dev,
                              int16_t* disp16_dev);

/* The batched form of sv_compute_rgb_host for callers whose images are already on the device (many
 * frames of one camera pair, e.g. a recorded stream): n_frames dense-packed RGB pairs
 * ([n][H][W][3] u8) -> [n][H][W] disparity16.  rectify != 0 runs the two cv::remap calls of
 * StereoCamera::rectifyImages (stereoCamera.cpp:573-574) with the context's maps on every frame;
 * then cvtColor x2 + the matcher (StereoMatcher.cpp:429-432), or with SV_FLAG_COST_RGB the colour
 * matcher directly (StereoMatcher.cpp:412).  rect_left_dev: NULL, or a caller-owned [n][H][W][3]
 * buffer that receives the rectified left images (the guide sv_filter_wls_dev wants).  n_frames <=
 * the sv_set_batch size.                                                                          */
int sv_compute_rgb_batch_dev(sv_ctx* ctx, int n_frames, const uint8_t* left_rgb_dev, const uint8_t* right_rgb_dev,
                             int rectify, uint8_t* rect_left_dev, int16_t* disp16_dev);

/* Replaces the two cv::cvtColor(.., CV_BGR2GRAY) calls of matchSGBMCUDA
 * (StereoMatcher.cpp:429-430) applied to RGB-ordered bytes; 3 bytes/pixel in,
 * 1 byte/pixel out, both on the device.                                       */
int sv_gray_from_rgb_dev(sv_ctx* ctx, const uint8_t* rgb_dev, uint8_t* gray_dev);

/* Replaces the map upload implicit in StereoCamera::rectifyImages
 * (stereoCamera.cpp:569-574): four CV_32FC1 maps (map11,map12,map21,map22),
 * width*height floats each, host memory.                                      */
int sv_upload_maps_host(sv_ctx* ctx, const float* map_lx, const float* map_ly,
                        const float* map_rx, const float* map_ry);

/* Replaces one cv::initUndistortRectifyMap(K, Dist, Rrect, Prect, size,
 * CV_32FC1, mapx, mapy) of rectifyImages (stereoCamera.cpp:569-570): the maps
 * of camera `which` are generated on the device (FP64, OpenCV's row-wise
 * running sums kept) and used by sv_remap_rgb_dev / sv_compute_rgb_host; no
 * map crosses PCIe.  The module re-runs this on every frame because the
 * kinematics mark the camera as changed (DispModule.cpp:684).  K 3x3, R 3x3,
 * P 3x4 row-major doubles (what cv::stereoRectify returned,
 * stereoCamera.cpp:564); dist = n_dist <= 8 coefficients k1 k2 p1 p2 k3 k4 k5
 * k6 (thin-prism / tilt terms are not modelled: SV_ERR_UNSUPPORTED).          */
int sv_init_rectify_map(sv_ctx* ctx, int which, const double* K, const double* dist, int n_dist,
                        const double* R, const double* P);

/* Replaces one cv::undistortPoints(grid, grid, K, Dist, Rrect, Prect) of
 * StereoCamera::updateMappings (stereoCamera.cpp:2437-2472): MapperL (which =
 * 0) / MapperR (1), CV_32FC2, kept on the device; sv_depth_* use MapperL when
 * called with a NULL mapper.                                                  */
int sv_update_mapper(sv_ctx* ctx, int which, const double* K, const double* dist, int n_dist,
                     const double* R, const double* P);

/* Read-back for callers that need the maps on the host (getMapperL,
 * stereoCamera.cpp:1915; DispModule::get3DPoints, DispModule.cpp:1103-1197). */
int sv_get_maps_host(sv_ctx* ctx, int which, float* mapx_host, float* mapy_host);
int sv_get_mapper_host(sv_ctx* ctx, int which, float* mapper_host);

/* Replaces one cv::remap(.., INTER_LINEAR) (BORDER_CONSTANT 0) of
 * rectifyImages (stereoCamera.cpp:573-574) on an 8UC3 image.  which = 0 left,
 * 1 right.  src/dst are device pointers, 3 bytes/pixel, dense.                */
int sv_remap_rgb_dev(sv_ctx* ctx, int which, const uint8_t* src_rgb_dev, uint8_t* dst_rgb_dev);

/* The whole of compute() for SGBM_HIP: remap both RGB images, convert to gray,
 * match.  RGB host images in, disparity16 host out; the rectified RGB pair is
 * kept on the device for the WLS guide and can be read back with
 * sv_get_rectified_host.                                                      */
int sv_compute_rgb_host(sv_ctx* ctx, const uint8_t* left_rgb_host, size_t left_stride,
                        const uint8_t* right_rgb_host, size_t right_stride, int rectify,
                        int16_t* disp16_host, size_t disp_stride);
/* The same plus the call that ends matchSGBMCUDA, getDisparityVis(disp16, vis, vis_scale)
 * (StereoMatcher.cpp:437): the 8-bit image is made on the device from the map that is still there and
 * comes back with it, one stream synchronisation for both.  vis_host may be NULL (= sv_compute_rgb_host). */
int sv_compute_rgb_vis_host(sv_ctx* ctx, const uint8_t* left_rgb_host, size_t left_stride,
                            const uint8_t* right_rgb_host, size_t right_stride, int rectify,
                            int16_t* disp16_host, size_t disp_stride, uint8_t* vis_host, size_t vis_stride, int vis_scale);
int sv_get_rectified_host(sv_ctx* ctx, int which, uint8_t* rgb_host, size_t stride);

/* Replaces cv::ximgproc::getDisparityVis(disp16, vis, 3)
 * (StereoMatcher.cpp:437): u8 = saturate(round_half_even(disp16*scale/16)).   */
int sv_disparity_vis_dev(sv_ctx* ctx, const int16_t* disp16_dev, uint8_t* vis_dev, int scale);
int sv_disparity_vis_batch_dev(sv_ctx* ctx, int n_frames, const int16_t* disp16_dev, uint8_t* vis_dev, int scale);
int sv_disparity_vis_host(sv_ctx* ctx, const int16_t* disp16_host, uint8_t* vis_host, int scale);

/* Replaces cv::filterSpeckles inside cv::StereoSGBM::compute (called by
 * matchSGBM, StereoMatcher.cpp:412) on a device disparity16 image, in place.  */
int sv_filter_speckles_dev(sv_ctx* ctx, int16_t* disp16_dev, int max_size, int max_diff);

/* The same filter for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO counterpart in the reference.  Same result as
 * sv_filter_speckles_dev, bit for bit, but the union-find labels and component sizes live in a
 * file-scope __device__ array of the library (2 ints per pixel for frames of up to
 * SV_STATIC_MAX_PIXELS pixels = 16.7 MB per GPU, held until the process ends) instead of buffers the
 * context allocates on first use: the call makes no hipMalloc / hipFree and no allocating library
 * call.  Larger frames are rejected with SV_ERR_UNSUPPORTED.  Every call on one GPU shares that GPU's
 * array (each call initialises what it uses), so two such calls that overlap in time on the same GPU
 * - different contexts on different streams - race: serialise them.  Asynchronous on the context
 * stream like the other *_dev entry points.                                                        */
#define SV_STATIC_MAX_PIXELS (1920 * 1088)
int sv_filter_speckles_static_dev(sv_ctx* ctx, int16_t* disp16_dev, int max_size, int max_diff);

/* Replaces the two wls_filter->filter(...) calls of StereoMatcherNew::filterWLS
 * (StereoMatcher.cpp:154-160) on the filter made by
 * createDisparityWLSFilterGeneric(false) (StereoMatcher.cpp:265): outside the
 * ROI of computeROI2 (StereoMatcher.cpp:12-29; sv_wls_roi returns {x,y,w,h})
 * the output is -16 (0 in the 8-bit image), inside it is the fast global
 * smoother (3 iterations, lambda x0.25 each) of the input guided by the
 * rectified left RGB image; lambda / sigma come from sv_params.  The 16-bit
 * map and, when vis != NULL, its 8-bit visualisation are filtered together,
 * as are the n_frames dense-packed frames of a batch.
 * _host: guide_rgb_host == NULL uses the rectified left image kept on the
 * device by the last sv_compute_rgb_host.                                     */
int sv_wls_roi(const sv_ctx* ctx, int* roi_xywh);
int sv_filter_wls_dev(sv_ctx* ctx, int n_frames, const int16_t* disp16_dev, const uint8_t* vis_dev,
                      const uint8_t* guide_rgb_dev, int16_t* disp16_wls_dev, uint8_t* vis_wls_dev);
int sv_filter_wls_host(sv_ctx* ctx, const int16_t* disp16_host, size_t disp_stride, const uint8_t* vis_host,
                       size_t vis_stride, const uint8_t* guide_rgb_host, size_t guide_stride,
                       int16_t* disp16_wls_host, size_t out16_stride, uint8_t* vis_wls_host, size_t out8_stride);

/* Replaces the BLF_ORIGINAL branch of StereoMatcherNew::filterBLF
 * (StereoMatcher.cpp:90-99): cv_extend::bilateralFilter(input16,
 * disparity16_blf, sigmaColorBLF, sigmaSpaceBLF), the FP64 bilateral-grid
 * filter of modules/DisparityModule/fastBilateral.hpp:92-189, followed by
 * getDisparityVis(disparity16_blf, disparity_blf, 3) when vis_blf != NULL.
 * Computed in FP64 in the reference's order of operations: bit-exact against
 * the CPU.  disp16_blf may be the input buffer.  One frame per call; the call
 * reads the map's min/max back to size the grid (one stream synchronisation).
 * Grids above 2^27 cells (sigmas near 1 on a large map) are rejected with
 * SV_ERR_UNSUPPORTED.  BLF_DISABLED needs no call (the reference clones);
 * BLF_CUDA: sv_filter_blf_joint_* below.                                      */
int sv_filter_blf_dev(sv_ctx* ctx, const int16_t* disp16_dev, double sigma_color, double sigma_space,
                      int16_t* disp16_blf_dev, uint8_t* vis_blf_dev);
int sv_filter_blf_host(sv_ctx* ctx, const int16_t* disp16_host, size_t disp_stride, double sigma_color,
                       double sigma_space, int16_t* disp16_blf_host, size_t out16_stride,
                       uint8_t* vis_blf_host, size_t out8_stride);

/* Replaces the BLF_CUDA branch of StereoMatcherNew::filterBLF (StereoMatcher.cpp:101-116; filter made
 * at :292-301): cv::cuda::DisparityBilateralFilter(ndisp = numberOfDisparities, radius 7, iters 2) with
 * sigma_range = sigmaColorBLF on the CV_16S map, guided by the gray rectified left image, followed by
 * getDisparityVis(.,.,3) when vis_blf != NULL.  EXT item (OpenCV-CUDA is not in the tree): restated
 * from the published algorithm, edge_threshold 0.1 and max_disc_threshold 0.2 as in OpenCV.  OpenCV's
 * kernel updates the map in place while neighbouring threads read it (result depends on timing);
 * here every half-pass reads the map as it was before the half-pass (deterministic).
 * _dev: guide_gray_dev = 8-bit gray [H][W].  _host: guide_rgb_host = 8UC3 rectified left image (the
 * cvtColor of :103-104 runs on the device), NULL = the one kept by the last sv_compute_rgb_host.
 * disp16_blf may be the input buffer (_dev).                                                        */
int sv_filter_blf_joint_dev(sv_ctx* ctx, const int16_t* disp16_dev, const uint8_t* guide_gray_dev, int ndisp, int radius,
                            int iters, double sigma_range, int16_t* disp16_blf_dev, uint8_t* vis_blf_dev);
int sv_filter_blf_joint_host(sv_ctx* ctx, const int16_t* disp16_host, size_t disp_stride, const uint8_t* guide_rgb_host,
                             size_t guide_stride, int ndisp, int radius, int iters, double sigma_range,
                             int16_t* disp16_blf_host, size_t out16_stride, uint8_t* vis_blf_host, size_t out8_stride);

/* Replaces DispModule::depthFromDisparity (DispModule.cpp:320-366), with the
 * non-aliased read.  mapper = MapperL (CV_32FC2,
 * width*height*2 floats; NULL = the device-resident MapperL made by
 * sv_update_mapper), Q 4x4 and R 3x3 row-major doubles, host memory.          */
int sv_depth_host(sv_ctx* ctx, const int16_t* disp16_host, size_t disp_stride,
                  const float* mapper_host, const double* Q, const double* R, float* depth_host,
                  size_t depth_stride);
int sv_depth_dev(sv_ctx* ctx, const int16_t* disp16_dev, const float* mapper_dev, const double* Q,
                 const double* R, float* depth_dev);
/* n_frames dense-packed disparity maps of one camera pair (same mapper, Q, R) in one launch */
int sv_depth_batch_dev(sv_ctx* ctx, int n_frames, const int16_t* disp16_dev, const float* mapper_dev,
                       const double* Q, const double* R, float* depth_dev);

/* Replaces the per-pixel loop of SFM::fillWorld3D (modules/SFM/SFM.cpp:
 * 1284-1347), i.e. DispModule::get3DPoints(u, v, "ROOT") (DispModule.cpp:
 * 1103-1160) for every pixel of the ORIGINAL left view: the cartesian image
 * (x, y, z in the root frame) and the cylindrical one (sqrt(x^2+y^2),
 * atan2(y, x), z), each width*height*3 floats (ImageOf<PixelRgbFloat>).
 * Pixels that map outside the rectified image or whose z is outside [0, 10] m
 * are 0.  mapper as in sv_depth_*; Q 4x4, RLrect 3x3 (not transposed: the
 * call transposes it as the reference does), HL_root 4x4, row-major doubles in
 * host memory.  FP64 like the reference; the disparity is read as the int16 it
 * is (SFM.cpp:1321 reads it with at<long>, a defect that is not reproduced).  */
int sv_world_image_dev(sv_ctx* ctx, const int16_t* disp16_dev, const float* mapper_dev, const double* Q,
                       const double* RLrect, const double* HL_root, float* cart_dev, float* cyl_dev);
int sv_world_image_host(sv_ctx* ctx, const int16_t* disp16_host, size_t disp_stride, const float* mapper_host,
                        const double* Q, const double* RLrect, const double* HL_root, float* cart_host,
                        size_t cart_stride, float* cyl_host, size_t cyl_stride);

/* Stage access for parity tests and profiling (no counterpart in the
 * reference beyond the #if 0 dump blocks, bova_sgm.cu:141-321).  Copies the
 * internal [H][W1][D] int16 volumes (W1 = computable columns = W - D for min_disparity 0) of the LAST sv_compute_* call to host.
 * which: 0 = C (matching cost + P2), 1 = S (aggregated).  S is only complete in
 * debug mode.  sv_set_debug modes: 0 = production (batches of >= 6 frames fuse
 * WTA into the last path, which then keeps the finished S in registers and
 * never stores it; smaller batches, being latency-bound, run WTA as its own
 * kernel; batches large enough to stream more than one frame through every
 * workgroup pipeline - about 48 frames at 640x480 on 256 CUs - with D <= 128
 * and all four paths of a sweep enabled run the fused sweep kernels, four paths
 * per pass over C),
 * 1 = debug (per-path kernels, S always materialised), 2 = WTA fused into the
 * last path whatever the batch size, 3 = fused sweeps whatever the batch size,
 * 4 = plain per-path launches (no fused sweeps, no concurrent paths), 5 = the
 * concurrent-path variant (every path on its own stream into its own volume,
 * then one sum + WTA kernel; what mode 0 picks for batches that leave most of
 * the chip idle, up to 8 frames at 640x480) whatever the batch size.  Modes
 * 2-5 let tests exercise each variant on one frame.                           */
int sv_set_debug(sv_ctx* ctx, int mode);
int sv_debug_read_volume_host(sv_ctx* ctx, int which, int16_t* vol_host, size_t n_elems);
/* which: 0 = disp before LR check, 1 = raw integer disparity, 2 = minS,
 * 3 = disp after LR check (before median).  width*height int16.               */
int sv_debug_read_plane_host(sv_ctx* ctx, int which, int16_t* plane_host);

/* Per-stage GPU time in milliseconds, measured with HIP events recorded on the
 * context stream around each stage of every sv_compute_* call made since
 * sv_enable_timing(ctx, 1); sv_get_timings synchronises the stream and returns
 * the MEAN over those calls (the most recent 64 at most).  For a batched call
 * the figures are per launch, i.e. for all frames of the batch.
 * Order: 0 prefilter, 1 cost, 2 aggregation, 3 wta, 4 lr+median(+speckle),
 * 5 total.  Replaces the commented PROF_S/PROF_E macros (DispModule.cpp:24-69). */
#define SV_N_TIMINGS 6
int sv_enable_timing(sv_ctx* ctx, int enable);
int sv_get_timings(sv_ctx* ctx, float* ms, int n);

/* Duration in milliseconds of each kernel launch of the aggregation stage (stage 2 above), in launch
 * order, from HIP events recorded on the context stream after every such launch: with the fused sweeps
 * ms[0] = first sweep (paths R, DR, D, DL), ms[1] = second sweep (paths L, UL, U, UR + winner-take-all);
 * with per-path kernels one entry per path.  Mean over the timed calls (the most recent 64 at most) that
 * launched as many kernels as the last one; *n_launches = that number (at most 8); entries past it are 0.
 * For bench.py's roofline line; no reference counterpart.                                        */
int sv_get_aggregation_timings(sv_ctx* ctx, float* ms, int n, int* n_launches);

/* How the last sv_compute_* call aggregated: kernel launches of the aggregation
 * stage, and how many of them were fused four-path sweeps (the rest are
 * single-path launches).  For profiling / bench.py; no reference counterpart. */
int sv_last_aggregation(const sv_ctx* ctx, int* n_launches, int* n_fused_sweeps);

/* Compute waves per workgroup (15 or 7; + one service wave) of the most recent fused-sweep launch, 0 if none
 * ran yet: with num_disparities / 32 it names the kernel instance, svk::k_sweep<N, second, waves>, in a
 * rocprofv3 trace.  For bench.py; no reference counterpart.                                       */
int sv_last_sweep_waves(const sv_ctx* ctx);

/* Diagnosis of the one assumption the fused sweeps make about the runtime when a launch is not fully resident
 * (sv_sweep.hip.h): workgroup i is given to XCD i mod 8, and every XCD starts its workgroups in blockIdx order (a band
 * waits for the band above it, which has the next lower blockIdx and therefore sits on the previous XCD).  With
 * enable != 0 every later sweep launch records the start time of each workgroup; the call returns, for the launches
 * recorded since the previous call: their number; the number of workgroups that started more than 20 us BEFORE workgroup
 * blockIdx - 8, their predecessor on the same XCD (expected: 0); the largest such lead in ticks of the 100 MHz
 * s_memrealtime counter (<= 0: none); and, for information, the largest lead over workgroup blockIdx - 1, which is on
 * another XCD (100+ us in an oversubscribed launch is normal).  It synchronises the context's stream.  A runtime that
 * breaks the rule is thereby named, not just timed out (SV_ERR_HIP from sv_synchronize after 2 s).  No reference
 * counterpart (bova_sgm.cu:108-127 launches independent blocks). */
int sv_debug_sweep_order(sv_ctx* ctx, int enable, int* launches, int* violations, long long* worst_lead_ticks,
                         long long* worst_lead_prev_ticks);

/* Self-test of the one floating-point instruction the integer path uses (sv_sweep.hip.h, pk_min3): the fused sweeps take their
 * three-input minima with gfx950's packed binary16 minimum, which equals the integer minimum on u16 values below 0x7C00 as long
 * as the kernels keep the default f16 denormal mode.  Runs this library's own build of that function over every pair of
 * [0, 0x7C00)^2 with a hashed third operand, both halves of the packed word (about 2e9 minima, a few ms), and returns the number
 * that differ from the integer minimum (expected: 0).  Synchronises the context's stream.  No reference counterpart. */
int sv_debug_selftest_min3(sv_ctx* ctx, unsigned long long* mismatches);

/* Algorithmic bytes per frame of the current geometry:
 * which: 0 cost, 1 aggregation, 2 wta, 3 tail, 4 frame.                       */
int64_t sv_algorithmic_bytes(const sv_ctx* ctx, int which);

const char* sv_last_error(const sv_ctx* ctx);
const char* sv_version(void);

/* In-process multi-device dispatch (BASELINE.json configs[4]): one context, one stream and - inside
 * sv_pool_compute_gray_host - one host thread per entry of `devices` (a device may be listed more than once: several
 * contexts share it).  Stereo pairs are independent: the n_frames pairs of a call are cut into chunks of `batch`
 * consecutive pairs and chunk k goes to context k mod n_contexts (sv_pool_plan states the rule and is what the call
 * follows); each context runs its chunks through sv_submit_gray_batch_host / sv_wait_batch_host with two chunks in
 * flight, so uploads, kernels and downloads overlap per device.  No collective and no traffic between devices; every
 * pair has its own output pointer, so results are in submission order by construction.  The call returns when every
 * pair is done (or with the first error; sv_pool_last_error).  Buffers as for sv_submit_gray_batch_host (pinned
 * memory makes the copies asynchronous).  One caller at a time per pool: the call is not reentrant (its contexts are
 * single-threaded-per-context like every sv_ctx); different pools may run concurrently.  A context whose host thread
 * cannot be started is served by the calling thread instead.  The reference handles one pair per updateModule() on one GPU
 * (DispModule.cpp:661-814, bova_sgm.cu:7-30); no counterpart there.                                           */
typedef struct sv_pool sv_pool;
int sv_pool_create(sv_pool** out, const int* devices, int n_devices, int width, int height, const sv_params* params, int batch);
int sv_pool_destroy(sv_pool* pool);
int sv_pool_size(const sv_pool* pool);
int sv_pool_plan(int n_frames, int n_contexts, int batch, int frame, int* context_index, int* chunk_index);
int sv_pool_compute_gray_host(sv_pool* pool, int n_frames, const uint8_t* const* left_host, size_t left_stride,
                              const uint8_t* const* right_host, size_t right_stride, int16_t* const* disp16_host,
                              size_t disp_stride);
const char* sv_pool_last_error(const sv_pool* pool);

#ifdef __cplusplus
}
#endif
#endif /* SV_HIP_H */
