/*
 * sv_ref.c — CPU oracle for the stereo hot path.  TEST INFRASTRUCTURE ONLY; see
 * sv_ref.h for the "parity unpinned" statement and for who may call this.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/modules/DisparityModule/ unless stated).  "REPO" = read from
 * the in-tree CUDA statement of the arithmetic; "EXT" = OpenCV behaviour the
 * reference relies on but does not contain.
 */
#include "sv_ref.h"

#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define SHRT_MAX_I 32767

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------------------------------------
 * A16 filterBLF, BLF_ORIGINAL branch (StereoMatcher.cpp:90-99): cv_extend::bilateralFilter on the
 * 16-bit disparity map, the bilateral-grid filter of modules/DisparityModule/fastBilateral.hpp:92-189
 * (REPO item: the whole algorithm is in the reference tree, in FP64).  Grid cells hold (sum, count);
 * here as two planes instead of the reference's interleaved pairs, the arithmetic per component is
 * the same.  Steps, with the reference's line numbers:
 *   :120-124 value range and grid size (padding 2 on every side),
 *   :131-143 splat every pixel into its nearest cell,
 *   :153-170 [1 2 1]/4 along y, y, x, x, z, z over the interior cells (border cells are never
 *            written and never receive a splat, so they stay 0),
 *   :174-177 sum /= count (count 0 -> divide by 1),
 *   :179-187 trilinear read-back at the pixel's own (y, x, value) position, index + 1 clamped,
 *   :108-109 convertTo(CV_16S): round half to even, saturate.
 * Returns 0, or -1 for non-positive sigmas / an allocation failure.
 * ------------------------------------------------------------------------------------------- */
static size_t blf_clampi(size_t lo, size_t hi, size_t x) { return x < lo ? lo : (x < hi ? x : hi); }

int svref_blf(const int16_t* src, int W, int H, double sigma_color, double sigma_space, int16_t* dst) {
    if (!(sigma_color > 0.0) || !(sigma_space > 0.0) || W <= 0 || H <= 0) return -1;
    const size_t n = (size_t)W * H, pad = 2;
    double vmin = src[0], vmax = src[0];
    for (size_t i = 1; i < n; i++) {
        if (src[i] < vmin) vmin = src[i];
        if (src[i] > vmax) vmax = src[i];
    }
    const size_t gh = (size_t)((double)(size_t)(H - 1) / sigma_space) + 1 + 2 * pad;
    const size_t gw = (size_t)((double)(size_t)(W - 1) / sigma_space) + 1 + 2 * pad;
    const size_t gd = (size_t)((vmax - vmin) / sigma_color) + 1 + 2 * pad;
    const size_t cells = gh * gw * gd;
    double* g[4]; /* sum, count and their ping-pong partners */
    for (int i = 0; i < 4; i++) {
        g[i] = (double*)calloc(cells, sizeof(double));
        if (!g[i]) { for (int j = 0; j < i; j++) free(g[j]); return -1; }
    }
    double *ds = g[0], *dc = g[1], *bs = g[2], *bc = g[3];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const double v = src[(size_t)y * W + x];
            const size_t sx = (size_t)((double)x / sigma_space + 0.5) + pad;
            const size_t sy = (size_t)((double)y / sigma_space + 0.5) + pad;
            const size_t sz = (size_t)((v - vmin) / sigma_color + 0.5) + pad;
            const size_t c = (sy * gw + sx) * gd + sz;
            ds[c] += v;
            dc[c] += 1.0;
        }
    const ptrdiff_t offs[3] = {(ptrdiff_t)(gw * gd), (ptrdiff_t)gd, 1};
    for (int dim = 0; dim < 3; dim++)
        for (int it = 0; it < 2; it++) {
            double* t;
            t = ds; ds = bs; bs = t;
            t = dc; dc = bc; bc = t;
            const ptrdiff_t off = offs[dim];
            for (size_t y = 1; y + 1 < gh; y++)
                for (size_t x = 1; x + 1 < gw; x++)
                    for (size_t z = 1; z + 1 < gd; z++) {
                        const size_t c = (y * gw + x) * gd + z;
                        ds[c] = (bs[c - off] + bs[c + off] + 2.0 * bs[c]) / 4.0;
                        dc[c] = (bc[c - off] + bc[c + off] + 2.0 * bc[c]) / 4.0;
                    }
        }
    for (size_t c = 0; c < cells; c++) ds[c] /= dc[c] != 0 ? dc[c] : 1;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const double z = (double)src[(size_t)y * W + x] - vmin;
            const double px = (double)x / sigma_space + (double)pad;
            const double py = (double)y / sigma_space + (double)pad;
            const double pz = z / sigma_color + (double)pad;
            const size_t y0 = blf_clampi(0, gh - 1, (size_t)py), y1 = blf_clampi(0, gh - 1, y0 + 1);
            const size_t x0 = blf_clampi(0, gw - 1, (size_t)px), x1 = blf_clampi(0, gw - 1, x0 + 1);
            const size_t z0 = blf_clampi(0, gd - 1, (size_t)pz), z1 = blf_clampi(0, gd - 1, z0 + 1);
            const double ya = py - (double)y0, xa = px - (double)x0, za = pz - (double)z0;
#define G(yy, xx, zz) ds[((yy) * gw + (xx)) * gd + (zz)]
            const double r = (1.0 - ya) * (1.0 - xa) * (1.0 - za) * G(y0, x0, z0) + (1.0 - ya) * xa * (1.0 - za) * G(y0, x1, z0) +
                             ya * (1.0 - xa) * (1.0 - za) * G(y1, x0, z0) + ya * xa * (1.0 - za) * G(y1, x1, z0) +
                             (1.0 - ya) * (1.0 - xa) * za * G(y0, x0, z1) + (1.0 - ya) * xa * za * G(y0, x1, z1) +
                             ya * (1.0 - xa) * za * G(y1, x0, z1) + ya * xa * za * G(y1, x1, z1);
#undef G
            const double q = nearbyint(r); /* cvRound: ties to even under the default rounding mode */
            dst[(size_t)y * W + x] = (int16_t)(q < -32768.0 ? -32768.0 : (q > 32767.0 ? 32767.0 : q));
        }
    for (int i = 0; i < 4; i++) free(g[i]);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * the world image SFM::fillWorld3D builds (modules/SFM/SFM.cpp:1284-1347), with
 * the per-point arithmetic of DispModule::get3DPoints, drive "ROOT" (DispModule.cpp:1103-1160).
 * For every ORIGINAL left pixel (u, v): (u', v') = MapperL(u, v), rounded half to even; outside the
 * rectified image -> the pixel stays 0.  d = disp16(v', u') / 16, w = d*Q32 + Q33,
 * p = ((u'+1)*Q00 + Q03, (v'+1)*Q11 + Q13, Q23) / w (doubles); z > 10 or z < 0 -> stays 0;
 * P = HL_root * [RLrect^T 0; 0 1] * (p, 1) (no division by P[3], SFM.cpp:1332);
 * cart = (x, y, z), cyl = (sqrt(x^2 + y^2), atan2(y, x), z), both as float.
 * Known defect NOT reproduced: SFM.cpp:1321 reads the CV_16SC1 map with at<long> (8 bytes at an
 * 8-byte-scaled offset); get3DPoints reads the same quantity correctly (cvGet2D, :1129) and that
 * is what is restated.  cart / cyl: [H][W][3] float.
 * ------------------------------------------------------------------------------------------- */
void svref_world_image(const int16_t* disp16, int W, int H, const float* mapper, const double* Q, const double* RLrect,
                       const double* HL_root, float* cart, float* cyl) {
    double Hr[16]; /* HL_root * [RLrect^T 0; 0 1] */
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double a = 0.0;
            for (int k = 0; k < 4; k++) {
                const double m = (k < 3 && j < 3) ? RLrect[j * 3 + k] : (k == j ? 1.0 : 0.0); /* transposed */
                a += HL_root[i * 4 + k] * m;
            }
            Hr[i * 4 + j] = a;
        }
    memset(cart, 0, (size_t)W * H * 3 * sizeof(float));
    memset(cyl, 0, (size_t)W * H * 3 * sizeof(float));
    for (int v = 0; v < H; v++)
        for (int u = 0; u < W; u++) {
            const size_t o = (size_t)v * W + u;
            const float usign = mapper[2 * o], vsign = mapper[2 * o + 1];
            const long u_ = lrintf(usign), v_ = lrintf(vsign);
            if (u_ < 0 || u_ >= W || v_ < 0 || v_ >= H) continue;
            double x = ((double)usign + 1) * Q[0] + Q[3];
            double y = ((double)vsign + 1) * Q[5] + Q[7];
            double z = Q[11];
            const double disparity = (double)disp16[(size_t)v_ * W + u_] / 16.0;
            const double w = disparity * Q[14] + Q[15];
            x /= w; y /= w; z /= w;
            if (z > 10.0 || z < 0.0) continue;
            const double X = Hr[0] * x + Hr[1] * y + Hr[2] * z + Hr[3];
            const double Y = Hr[4] * x + Hr[5] * y + Hr[6] * z + Hr[7];
            const double Z = Hr[8] * x + Hr[9] * y + Hr[10] * z + Hr[11];
            cart[3 * o] = (float)X; cart[3 * o + 1] = (float)Y; cart[3 * o + 2] = (float)Z;
            cyl[3 * o] = (float)sqrt(X * X + Y * Y); cyl[3 * o + 1] = (float)atan2(Y, X); cyl[3 * o + 2] = (float)Z;
        }
}

/* ---------------------------------------------------------------------------------------------
 * A16 filterBLF, BLF_CUDA branch (StereoMatcher.cpp:101-116, set up at :292-301): EXT item,
 * cv::cuda::DisparityBilateralFilter(ndisp = numberOfDisparities, radius 7, iters 2) with
 * sigma_range = sigmaColorBLF, edge_threshold 0.1, max_disc_threshold 0.2 (its defaults), applied
 * to the CV_16S disparity16 map with the gray rectified left image as the guide.  Restated from
 * the published algorithm (OpenCV cudastereo, disp_bilateral_filter; OpenCV is not installed:
 * parity unpinned):
 *   table_color[i] = (float)exp(-(double)(i*i) / (2 sigma^2)),      i = |guide difference|
 *   table_space[dy][dx] = expf(-sqrtf(dy*dy + dx*dx) / (radius + 1))
 *   edge_disc = max(1, (short)(ndisp*0.1 + 0.5)),  max_disc = (short)(ndisp*0.2 + 0.5)
 *   iters x { half-pass t = 0, half-pass t = 1 }; a half-pass visits the interior pixels with
 *   (x + y + t) even... precisely x = 2k + ((y + t) & 1); a pixel whose 4-neighbourhood differs from
 *   it by >= edge_disc takes, among {itself, up, left, down, right}, the value with the smallest
 *   cost = sum over the (2r+1)^2 window (clipped to the image, rows then columns ascending) of
 *          min(max_disc, |disp(window) - candidate|) * table_color * table_space   (float adds);
 *   ties go to the first candidate in that order.
 * OpenCV's kernel updates the map in place while other threads still read it, so its result depends
 * on thread timing; this restatement fixes the schedule "every pixel of a half-pass reads the map as
 * it was before the half-pass".  OpenCV's colour table has 255 entries and is indexed with up to
 * 255; here it has 256.
 * ------------------------------------------------------------------------------------------- */
int svref_blf_joint(const int16_t* disp, const uint8_t* guide, int W, int H, int ndisp, int radius, int iters, double sigma_range,
                    int16_t* out) {
    if (W < 1 || H < 1 || ndisp < 1 || radius < 1 || radius > 32 || iters < 1 || !(sigma_range > 0.0)) return -1;
    float tc[256], ts[33][33];
    for (int i = 0; i < 256; i++) tc[i] = (float)exp(-(double)(i * i) / (2 * sigma_range * sigma_range));
    const float dist_space = (float)radius + 1.0f;
    for (int y = 0; y <= radius; y++)
        for (int x = 0; x <= radius; x++) ts[y][x] = expf(-sqrtf((float)(y * y) + (float)(x * x)) / dist_space);
    short edge_disc = (short)(ndisp * 0.1f + 0.5);
    if (edge_disc < 1) edge_disc = 1;
    const short max_disc = (short)(ndisp * 0.2f + 0.5);
    const size_t n = (size_t)W * H;
    int16_t* cur = (int16_t*)malloc(n * sizeof(int16_t));
    if (!cur) return -1;
    memcpy(out, disp, n * sizeof(int16_t));
    for (int it = 0; it < iters; it++)
        for (int t = 0; t < 2; t++) {
            memcpy(cur, out, n * sizeof(int16_t)); /* the map as it was before this half-pass */
            for (int y = 1; y < H - 1; y++)
                for (int x = 1 + ((y + t + 1) & 1); x < W - 1; x += 2) { /* x = 2k + ((y + t) & 1), x >= 1 */
                    int dp[5];
                    dp[0] = cur[(size_t)y * W + x];
                    dp[1] = cur[(size_t)(y - 1) * W + x];
                    dp[2] = cur[(size_t)y * W + x - 1];
                    dp[3] = cur[(size_t)(y + 1) * W + x];
                    dp[4] = cur[(size_t)y * W + x + 1];
                    if (!(abs(dp[1] - dp[0]) >= edge_disc || abs(dp[2] - dp[0]) >= edge_disc || abs(dp[3] - dp[0]) >= edge_disc ||
                          abs(dp[4] - dp[0]) >= edge_disc))
                        continue;
                    const int ymin = imax(0, y - radius), xmin = imax(0, x - radius);
                    const int ymax = imin(H - 1, y + radius), xmax = imin(W - 1, x + radius);
                    float cost[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    const int ic = guide[(size_t)y * W + x];
                    for (int yi = ymin; yi <= ymax; yi++)
                        for (int xi = xmin; xi <= xmax; xi++) {
                            const float weight = tc[abs((int)guide[(size_t)yi * W + xi] - ic)] * ts[abs(y - yi)][abs(x - xi)];
                            const int dr = cur[(size_t)yi * W + xi];
                            for (int k = 0; k < 5; k++) cost[k] += (float)imin(max_disc, abs(dr - dp[k])) * weight;
                        }
                    float minimum = 3.402823466e+38f;
                    int id = 0;
                    for (int k = 0; k < 5; k++)
                        if (cost[k] < minimum) { minimum = cost[k]; id = k; }
                    out[(size_t)y * W + x] = (int16_t)dp[id];
                }
        }
    free(cur);
    return 0;
}

const char* svref_version(void) { return "sv_ref 1 (restated SGBM, parity unpinned)"; }

/* min_disparity (EXT: cv::StereoSGBM, as matchSGBM passes it, StereoMatcher.cpp:399; the CUDA back-end has no
 * such parameter).  OpenCV's computeDisparitySGBM: minD = minDisparity, maxD = minD + numDisparities; the
 * computable columns are x in [minX1, maxX1) = [max(maxD, 0), W + min(minD, 0)); cost index d (0..D-1) pairs
 * left x with right x - d - minD; the stored disparity is (d + minD) * 16 + sub-pixel; every other pixel is
 * INVALID_DISP_SCALED = (minD - 1) * 16. */
static int geo_col0(int W, int D, int minD) { (void)W; return imax(minD + D, 0); }
static int geo_colN(int W, int D, int minD) { (void)D; return W + imin(minD, 0); }
static int geo_w1(int W, int D, int minD) { return geo_colN(W, D, minD) - geo_col0(W, D, minD); }
static int geo_invalid(int minD) { return (minD - 1) * 16; }

/* Limits under which every int16 store of the reference is wrap-free, so the
 * restatement (and the GPU) may compute windows directly instead of through
 * the reference's running sums.  Even block sizes are rejected (Appendix A9b:
 * REPO and EXT disagree there). */
int svref_check_params(const svref_params* p, int W, int H) {
    if (!p) return -1;
    int D = p->num_disparities, BS = p->block_size;
    if (p->min_disparity < -1024 || p->min_disparity > 1024) return -2; /* (minD - 1) * 16 and (minD + D) * 16 fit int16 */
    if (D < 32 || D > 256 || (D % 32) != 0) return -2;
    if (BS < 1 || BS > 15 || (BS % 2) == 0) return -2;   /* the GUI offers 3..31 (cvgui.cpp:100); beyond 15 no cap leaves C inside int16 */
    if (H < BS || geo_w1(W, D, p->min_disparity) < BS || geo_w1(W, D, p->min_disparity) < 1) return -1;
    if (p->pre_filter_cap < 0 || p->pre_filter_cap > 127) return -2;   /* clip(v) + cap is stored in a PixType (u8, costs.cu:286-292): 2 * cap <= 255; the GUI offers 0..100 (cvgui.cpp:102) */
    if (p->P1 < 0 || p->P2 < p->P1 || p->P2 < 1) return -2;
    /* max per-pixel BT cost: 2*cap (gradient channel) + 255>>2 (raw channel), per colour channel */
    const int cn = (p->flags & SVREF_FLAG_COST_RGB) ? 3 : 1;
    long cmax = (long)cn * (2 * p->pre_filter_cap + 63) * BS * BS + p->P2;
    if (cmax > 32767) return -2;
    if ((long)p->P1 + 32767 > 65535) return -2;
    if (p->uniqueness_ratio < 0 || p->uniqueness_ratio > 100) return -2;
    if ((p->path_mask & 0xFF) == 0 || (p->path_mask & ~0xFFu)) return -2;
    return 0;
}

/* A1 REPO costs.cu:116-166 (get_gradient) with the clip table of
 * costs.cu:286-292 (fill_tab; TAB_OFS bova_sgm.cu:31) folded into clip().
 * grad is computed from the UNMODIFIED image (the __syncthreads at :151 orders
 * the phases); afterwards column 0 and W-1 of grad and of the raw channel are
 * forced to tab[0] = ftzero (:153-164). */
void svref_prefilter(const uint8_t* img, int W, int H, int cap, uint8_t* grad, uint8_t* raw_mod) {
    const int ftzero = cap;
    for (int y = 0; y < H; y++) {
        const uint8_t* r0 = img + (size_t)y * W;
        const uint8_t* rm = img + (size_t)imax(y - 1, 0) * W;
        const uint8_t* rp = img + (size_t)imin(y + 1, H - 1) * W;
        uint8_t* g = grad + (size_t)y * W;
        uint8_t* rw = raw_mod + (size_t)y * W;
        for (int x = 1; x < W - 1; x++) {
            int v = (r0[x + 1] - r0[x - 1]) * 2 + rm[x + 1] - rm[x - 1] + rp[x + 1] - rp[x - 1];
            g[x] = (uint8_t)(imin(imax(v, -ftzero), ftzero) + ftzero);
            rw[x] = r0[x];
        }
        g[0] = g[W - 1] = (uint8_t)ftzero;
        rw[0] = rw[W - 1] = (uint8_t)ftzero;
    }
}

/* A2 REPO costs.cu:170-201 (get_pixel_diff), one channel, accumulated into
 * pix[y][xl][d] (int16 store at :199). */
static void bt_channel(const uint8_t* L, const uint8_t* R, int W, int H, int D, int minD, int shift, int16_t* pix) {
    const int col0 = geo_col0(W, D, minD), colN = geo_colN(W, D, minD), W1 = colN - col0;
    for (int y = 0; y < H; y++) {
        const uint8_t* l = L + (size_t)y * W;
        const uint8_t* r = R + (size_t)y * W;
        for (int x = col0; x < colN; x++) {
            int u = l[x];
            int ul = x > 0 ? (u + l[x - 1]) / 2 : u;
            int ur = x < W - 1 ? (u + l[x + 1]) / 2 : u;
            int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
            int16_t* out = pix + ((size_t)y * W1 + (x - col0)) * D;
            for (int d = 0; d < D; d++) {
                const int xr = x - d - minD; /* in [1, W-1] for every computable x */
                int v = r[xr];
                int vl = xr >= 1 ? (r[xr - 1] + v) / 2 : v;
                int vr = xr < W - 1 ? (r[xr + 1] + v) / 2 : v;
                int v0 = imin(imin(vl, vr), v), v1 = imax(imax(vl, vr), v);
                int c0 = imax(imax(0, u - v1), v0 - u);
                int c1 = imax(imax(0, v - u1), u0 - v);
                out[d] = (int16_t)(out[d] + (imin(c0, c1) >> shift));
            }
        }
    }
}

/* A2-A4.  REPO: bova_sgm.cu:94-103 (stage order), costs.cu:203-233 (get_hsum),
 * costs.cu:235-284 (get_cost).  Volumes are [H][W1][D], d innermost, over the
 * computable columns x in [D, W) only. */
static int cost_volume_planes(const uint8_t* const* lefts, const uint8_t* const* rights, int cn, int W, int H,
                              const svref_params* p, int16_t* C, svref_stages* st);

int svref_cost_volume(const uint8_t* left, const uint8_t* right, int W, int H, const svref_params* p,
                      int16_t* C, svref_stages* st) {
    return cost_volume_planes(&left, &right, 1, W, H, p, C, st);
}

/* cn = 1: the gray path of the CUDA back-end (matchSGBMCUDA).  cn = 3 (EXT): what cv::StereoSGBM does
 * with the 8UC3 images matchSGBM hands it (StereoMatcher.cpp:392-417): calcPixelCostBT runs the same
 * prefilter + BT procedure on every colour plane - cn gradient rows with diff_scale 0, then cn raw
 * rows with diff_scale 2, edge columns of all 2*cn rows forced to tab[0] - and adds the 2*cn costs
 * into one cost per (x, d).  Stage outputs grad / raw hold the LAST plane. */
static int cost_volume_planes(const uint8_t* const* lefts, const uint8_t* const* rights, int cn, int W, int H,
                              const svref_params* p, int16_t* C, svref_stages* st) {
    const int D = p->num_disparities, minD = p->min_disparity, W1 = geo_w1(W, D, minD), BS = p->block_size, SW2 = BS / 2, SH2 = BS / 2;
    const int P2 = p->P2;
    const size_t npx = (size_t)W * H, nvol = (size_t)W1 * H * D;
    uint8_t* gl = (uint8_t*)malloc(npx);
    uint8_t* gr = (uint8_t*)malloc(npx);
    uint8_t* rl = (uint8_t*)malloc(npx);
    uint8_t* rr = (uint8_t*)malloc(npx);
    int16_t* pix = (int16_t*)calloc(nvol, sizeof(int16_t)); /* cudaMemset, bova_sgm.cu:94 */
    int16_t* hs = (int16_t*)malloc(nvol * sizeof(int16_t));
    if (!gl || !gr || !rl || !rr || !pix || !hs) {
        free(gl); free(gr); free(rl); free(rr); free(pix); free(hs);
        return -6;
    }
    for (int ch = 0; ch < cn; ch++) {
        svref_prefilter(lefts[ch], W, H, p->pre_filter_cap, gl, rl);
        svref_prefilter(rights[ch], W, H, p->pre_filter_cap, gr, rr);
        bt_channel(gl, gr, W, H, D, minD, 0, pix); /* bova_sgm.cu:95 */
        bt_channel(rl, rr, W, H, D, minD, 2, pix); /* bova_sgm.cu:97 */
    }

    /* A3 horizontal box, costs.cu:214-232 */
    for (int y = 0; y < H; y++) {
        const int16_t* px = pix + (size_t)y * W1 * D;
        int16_t* h = hs + (size_t)y * W1 * D;
        for (int d = 0; d < D; d++) {
            int acc = 0;
            for (int k = 0; k <= SW2; k++) acc += px[(size_t)k * D + d] * (k == 0 ? SW2 + 1 : 1);
            h[d] = (int16_t)acc;
        }
        for (int xl = 1; xl < W1; xl++) {
            const int16_t* add = px + (size_t)imin(xl + SW2, W1 - 1) * D;
            const int16_t* sub = px + (size_t)imax(xl - SW2 - 1, 0) * D;
            const int16_t* prev = h + (size_t)(xl - 1) * D;
            int16_t* cur = h + (size_t)xl * D;
            for (int d = 0; d < D; d++) cur[d] = (int16_t)(prev[d] + add[d] - sub[d]);
        }
    }

    /* A4 vertical box + P2, costs.cu:244-283 */
    const size_t rowN = (size_t)W1 * D;
    for (int xl = 0; xl < W1; xl++) {
        for (int d = 0; d < D; d++) {
            size_t o = (size_t)xl * D + d;
            int acc = P2;
            for (int i = 0; i <= SH2; i++) acc += hs[(size_t)i * rowN + o] * (i == 0 ? SH2 + 1 : 1);
            C[o] = (int16_t)acc;
        }
    }
    for (int y = 1; y < H; y++) {
        const int k = y + SH2;
        int16_t* c = C + (size_t)y * rowN;
        const int16_t* cp = C + (size_t)(y - 1) * rowN;
        if (k < H) {
            const int16_t* add = hs + (size_t)k * rowN;
            const int16_t* sub = hs + (size_t)(k >= BS ? k - BS : 0) * rowN;
            /* xl = 0: "copy opencv" quirk, costs.cu:254-267 */
            for (int d = 0; d < D; d++) c[d] = cp[d];
            for (size_t o = D; o < rowN; o++) c[o] = (int16_t)(cp[o] + add[o] - sub[o]);
        } else {
            /* bottom SH2 rows keep the previous row, costs.cu:263-266,279-282 */
            memcpy(c, cp, rowN * sizeof(int16_t));
        }
    }
    /* EXT MODE_HH alternatives (Appendix A4): rows own their C buffer there, so
     * what REPO leaves as a copy is the initial P2. */
    if (p->flags & SVREF_FLAG_C_TAIL_P2)
        for (int y = imax(H - SH2, 1); y < H; y++) {
            int16_t* c = C + (size_t)y * rowN;
            for (size_t o = 0; o < rowN; o++) c[o] = (int16_t)P2;
        }
    if (p->flags & SVREF_FLAG_C_COL0_P2)
        for (int y = 1; y < H; y++) {
            int16_t* c = C + (size_t)y * rowN;
            for (int d = 0; d < D; d++) c[d] = (int16_t)P2;
        }

    if (st) {
        if (st->grad_l) memcpy(st->grad_l, gl, npx);
        if (st->grad_r) memcpy(st->grad_r, gr, npx);
        if (st->raw_l) memcpy(st->raw_l, rl, npx);
        if (st->raw_r) memcpy(st->raw_r, rr, npx);
        if (st->pix) memcpy(st->pix, pix, nvol * sizeof(int16_t));
        if (st->hsum) memcpy(st->hsum, hs, nvol * sizeof(int16_t));
        if (st->C) memcpy(st->C, C, nvol * sizeof(int16_t));
    }
    free(gl); free(gr); free(rl); free(rr); free(pix); free(hs);
    return 0;
}

static const int DIR_DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DIR_DY[8] = {0, 1, 1, 1, 0, -1, -1, -1};

/* A5.  REPO aggregation.cu:60-141 (LR), :147-306 (UD_LR), :308-385 (UD),
 * :388-543 (RL_UD), :546-625 (RL); the three upward directions are the second
 * sweep of EXT MODE_HH with the identical recurrence.  A predecessor outside
 * [0,W1)x[0,H) is a fresh path start (Lp == 0, delta = P2): that is what the
 * kernels do at row 0 (:75-86) and what the diagonal kernels do when they wrap
 * at an image edge (:233-256, :469-492).  Unlike REPO (Appendix A5 caveat) every
 * pixel is covered whatever the aspect ratio.  safe_add: aggregation.cu:4-10. */
void svref_aggregate_path(const int16_t* C, int W1, int H, int D, int P1, int P2, int dir, int16_t* S,
                          int16_t* L_out) {
    const int dx = DIR_DX[dir], dy = DIR_DY[dir];
    const size_t rowN = (size_t)W1 * D;
    /* L rows carry guard entries at d = -1 and d = D (lr_pre[-1], lr_pre[D] = SHRT_MAX, :78) */
    const int DP = D + 2;
    int16_t* rows = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)W1 * DP);
    int* mins = (int*)malloc(sizeof(int) * 2 * (size_t)W1);
    int16_t* fresh = (int16_t*)calloc(DP, sizeof(int16_t));
    fresh[0] = fresh[D + 1] = SHRT_MAX_I;
    for (size_t i = 0; i < 2 * (size_t)W1; i++) rows[i * DP] = rows[i * DP + D + 1] = SHRT_MAX_I;

    for (int iy = 0; iy < H; iy++) {
        const int y = dy < 0 ? H - 1 - iy : iy;
        int16_t* cur = rows + (size_t)(iy & 1) * W1 * DP;
        int16_t* prv = rows + (size_t)((iy & 1) ^ 1) * W1 * DP;
        int* curmin = mins + (size_t)(iy & 1) * W1;
        int* prvmin = mins + (size_t)((iy & 1) ^ 1) * W1;
        const int py = y - dy;
        for (int ix = 0; ix < W1; ix++) {
            const int x = dx < 0 ? W1 - 1 - ix : ix;
            const int px = x - dx;
            const int has_pred = px >= 0 && px < W1 && py >= 0 && py < H;
            const int16_t* Lp = fresh + 1;
            int delta = P2;
            if (has_pred) {
                if (dy == 0) {
                    Lp = cur + (size_t)px * DP + 1;
                    delta = curmin[px] + P2;
                } else {
                    Lp = prv + (size_t)px * DP + 1;
                    delta = prvmin[px] + P2;
                }
            }
            const int16_t* c = C + (size_t)y * rowN + (size_t)x * D;
            int16_t* s = S + (size_t)y * rowN + (size_t)x * D;
            int16_t* Lc = cur + (size_t)x * DP + 1;
            int mn = SHRT_MAX_I;
            for (int d = 0; d < D; d++) {
                int m = imin(imin((int)Lp[d], Lp[d - 1] + P1), imin(Lp[d + 1] + P1, delta));
                int l = c[d] + m - delta;
                Lc[d] = (int16_t)l;
                mn = imin(mn, l);
                int t = s[d] + l;
                s[d] = (int16_t)(t > SHRT_MAX_I ? SHRT_MAX_I : t);
            }
            curmin[x] = mn;
            if (L_out) memcpy(L_out + (size_t)y * rowN + (size_t)x * D, Lc, sizeof(int16_t) * D);
        }
    }
    free(rows);
    free(mins);
    free(fresh);
}

/* A6.  REPO aggregation.cu:629-689 (get_disparity).  Full-width planes; columns
 * [0,D) get the contract value (Appendix A9).  raw = -1 marks "no disparity"
 * (REPO writes INVALID_DISP_SCALED there, :642; same meaning). */
void svref_wta(const int16_t* S, int W, int H, int D, int uniq, int16_t* disp, int16_t* raw, int16_t* mins) {
    svref_wta_min(S, W, H, D, 0, uniq, disp, raw, mins);
}
/* the same with min_disparity (EXT): raw stays the cost INDEX (0..D-1, -1 = none), disp carries minD */
void svref_wta_min(const int16_t* S, int W, int H, int D, int minD, int uniq, int16_t* disp, int16_t* raw, int16_t* mins) {
    const int col0 = geo_col0(W, D, minD), colN = geo_colN(W, D, minD), W1 = colN - col0;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            size_t o = (size_t)y * W + x;
            disp[o] = (int16_t)geo_invalid(minD);
            raw[o] = -1;
            mins[o] = SHRT_MAX_I;
            if (x < col0 || x >= colN) continue;
            const int16_t* s = S + ((size_t)y * W1 + (x - col0)) * D;
            int minS = SHRT_MAX_I, best = -1;
            for (int d = 0; d < D; d++)
                if (minS > s[d]) { minS = s[d]; best = d; }
            mins[o] = (int16_t)minS;
            if (best < 0) continue; /* every S saturated: no minimum, invalid */
            int d;
            for (d = 0; d < D; d++)
                if (s[d] * (100 - uniq) < minS * 100 && abs(best - d) > 1) break;
            if (d < D) continue;
            raw[o] = (int16_t)best;
            int v;
            if (0 < best && best < D - 1) {
                int denom2 = imax(s[best - 1] + s[best + 1] - 2 * s[best], 1);
                v = best * 16 + ((s[best - 1] - s[best + 1]) * 16 + denom2) / (denom2 * 2);
            } else
                v = best * 16;
            disp[o] = (int16_t)(v + minD * 16);
        }
    }
}

/* A7.  REPO aggregation.cu:691-735 (lrcheck).  disp2 is built right to left
 * with a strict '>' so ties keep the larger x (:706-719). */
void svref_lrcheck(int16_t* disp, const int16_t* raw, const int16_t* mins, int W, int H, int D, int thresh,
                   int16_t* disp2_out) {
    svref_lrcheck_min(disp, raw, mins, W, H, D, 0, thresh, disp2_out);
}
/* with min_disparity (EXT): disp2 holds true disparities (index + minD).  An entry nobody wrote keeps the marker
 * cv::StereoSGBM initialises it with, INVALID_DISP_SCALED = (minD - 1) * 16 - the SCALED marker in an unscaled
 * array - and its test "disp2 >= minD" (the reference shows the pattern for minD = 0, aggregation.cu:703,732) lets
 * that marker through as a disparity as soon as (minD - 1) * 16 >= minD, i.e. for every minD >= 2: an unwritten
 * right-image column then counts as disparity (minD - 1) * 16 in the consistency test.  Reproduced as is. */
void svref_lrcheck_min(int16_t* disp, const int16_t* raw, const int16_t* mins, int W, int H, int D, int minD, int thresh,
                       int16_t* disp2_out) {
    const int col0 = geo_col0(W, D, minD), colN = geo_colN(W, D, minD), inv = geo_invalid(minD);
    int16_t* disp2 = (int16_t*)malloc(sizeof(int16_t) * W);
    int16_t* d2c = (int16_t*)malloc(sizeof(int16_t) * W);
    for (int y = 0; y < H; y++) {
        int16_t* dp = disp + (size_t)y * W;
        const int16_t* rw = raw + (size_t)y * W;
        const int16_t* mn = mins + (size_t)y * W;
        for (int x = 0; x < W; x++) { d2c[x] = SHRT_MAX_I; disp2[x] = (int16_t)((minD - 1) * 16); }
        for (int x = colN - 1; x >= col0; x--) {
            int d = rw[x];
            if (d < 0) continue;
            int x2 = x - d - minD;
            if (d2c[x2] > mn[x]) { d2c[x2] = mn[x]; disp2[x2] = (int16_t)(d + minD); }
        }
        for (int x = col0; x < colN; x++) {
            int d1 = dp[x];
            if (d1 == inv) continue;
            int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < W && disp2[_x] >= minD && abs(disp2[_x] - _d) > thresh && 0 <= x_ && x_ < W &&
                disp2[x_] >= minD && abs(disp2[x_] - d_) > thresh)
                dp[x] = (int16_t)inv;
        }
        if (disp2_out) memcpy(disp2_out + (size_t)y * W, disp2, sizeof(int16_t) * W);
    }
    free(disp2);
    free(d2c);
}

static int16_t median9(int16_t* w) {
    /* partial selection sort, as aggregation.cu:753-764 */
    for (int i = 0; i < 5; i++) {
        int m = i;
        for (int j = i + 1; j < 9; j++)
            if (w[j] < w[m]) m = j;
        int16_t t = w[i]; w[i] = w[m]; w[m] = t;
    }
    return w[4];
}

/* A8.  REPO aggregation.cu:737-768 (MedianFilter): interior 3x3 median, border
 * pixels copied.  replicate = 1: EXT cv::medianBlur(.,3) border replication. */
void svref_median3(const int16_t* in, int16_t* out, int W, int H, int replicate) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int border = (y < 1 || x < 1 || y >= H - 1 || x >= W - 1);
            if (border && !replicate) { out[(size_t)y * W + x] = in[(size_t)y * W + x]; continue; }
            int16_t w[9];
            int n = 0;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++) {
                    int yy = imin(imax(y + i, 0), H - 1), xx = imin(imax(x + j, 0), W - 1);
                    w[n++] = in[(size_t)yy * W + xx];
                }
            out[(size_t)y * W + x] = median9(w);
        }
}

/* A10.  EXT cv::filterSpeckles (called inside cv::StereoSGBM::compute, which
 * matchSGBM uses, StereoMatcher.cpp:412): 4-connected components over pixels
 * != new_val, neighbours joined iff |delta| <= max_diff; components of size
 * <= max_size are set to new_val.  Order-independent. */
void svref_filter_speckles(int16_t* img, int W, int H, int new_val, int max_size, int max_diff) {
    const size_t n = (size_t)W * H;
    int32_t* label = (int32_t*)calloc(n, sizeof(int32_t));
    int32_t* stack = (int32_t*)malloc(n * sizeof(int32_t));
    uint8_t* small = (uint8_t*)malloc(n + 1);
    int cur = 0;
    for (size_t i = 0; i < n; i++) {
        if (img[i] == new_val) continue;
        if (label[i]) {
            if (small[label[i]]) img[i] = (int16_t)new_val;
            continue;
        }
        int sp = 0, count = 0;
        cur++;
        label[i] = cur;
        stack[sp++] = (int32_t)i;
        while (sp) {
            int32_t p = stack[--sp];
            count++;
            int py = p / W, px = p % W, v = img[p];
            const int nb[4][2] = {{py, px + 1}, {py, px - 1}, {py + 1, px}, {py - 1, px}};
            for (int k = 0; k < 4; k++) {
                int yy = nb[k][0], xx = nb[k][1];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                int32_t q = yy * W + xx;
                if (label[q] || img[q] == new_val || abs(img[q] - v) > max_diff) continue;
                label[q] = cur;
                stack[sp++] = q;
            }
        }
        small[cur] = count <= max_size;
        if (small[cur]) img[i] = (int16_t)new_val;
    }
    free(label);
    free(stack);
    free(small);
}

/* Whole matcher.  Stage order: bova_sgm.cu:87-136; columns [0,D) = -16 (A9);
 * median then speckle as cv::StereoSGBM::compute does (EXT). */
static int sgbm_planes(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int cn, int W, int H,
                       const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st);

int svref_sgbm(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int W, int H,
               const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st) {
    if (p && (p->flags & SVREF_FLAG_COST_RGB)) return -2; /* the colour cost needs the RGB entry point */
    return sgbm_planes(left, ls, right, rs, 1, W, H, p, disp16, ds, st);
}

/* matchSGBM (StereoMatcher.cpp:392-417): StereoSGBM on the 8UC3 pair, interleaved RGB, strides in bytes */
int svref_sgbm_rgb(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int W, int H,
                   const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st) {
    if (!p || !(p->flags & SVREF_FLAG_COST_RGB)) return -2;
    return sgbm_planes(left, ls, right, rs, 3, W, H, p, disp16, ds, st);
}

static int sgbm_planes(const uint8_t* left, size_t ls, const uint8_t* right, size_t rs, int cn, int W, int H,
                       const svref_params* p, int16_t* disp16, size_t ds, svref_stages* st) {
    int rc = svref_check_params(p, W, H);
    if (rc) return rc;
    if (!left || !right || !disp16 || ls < (size_t)W * cn || rs < (size_t)W * cn || ds < (size_t)W * 2) return -1;
    const int D = p->num_disparities, minD = p->min_disparity, W1 = geo_w1(W, D, minD);
    const size_t npx = (size_t)W * H, nvol = (size_t)W1 * H * D;
    uint8_t* l = (uint8_t*)malloc(npx * cn); /* cn planes */
    uint8_t* r = (uint8_t*)malloc(npx * cn);
    int16_t* C = (int16_t*)malloc(nvol * sizeof(int16_t));
    int16_t* S = (int16_t*)calloc(nvol, sizeof(int16_t)); /* cudaMemset, bova_sgm.cu:106 */
    int16_t* disp = (int16_t*)malloc(npx * 2);
    int16_t* raw = (int16_t*)malloc(npx * 2);
    int16_t* mins = (int16_t*)malloc(npx * 2);
    int16_t* med = (int16_t*)malloc(npx * 2);
    if (!l || !r || !C || !S || !disp || !raw || !mins || !med) {
        free(l); free(r); free(C); free(S); free(disp); free(raw); free(mins); free(med);
        return -6;
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int ch = 0; ch < cn; ch++) {
                l[(size_t)ch * npx + (size_t)y * W + x] = left[y * ls + (size_t)x * cn + ch];
                r[(size_t)ch * npx + (size_t)y * W + x] = right[y * rs + (size_t)x * cn + ch];
            }
    const uint8_t* lp[3] = {l, l + npx, l + 2 * npx};
    const uint8_t* rp[3] = {r, r + npx, r + 2 * npx};
    rc = cost_volume_planes(lp, rp, cn, W, H, p, C, st);
    if (rc == 0) {
        for (int dir = 0; dir < 8; dir++)
            if (p->path_mask & (1u << dir))
                svref_aggregate_path(C, W1, H, D, p->P1, p->P2, dir, S, st ? st->L[dir] : NULL);
        if (st && st->S) memcpy(st->S, S, nvol * sizeof(int16_t));
        svref_wta_min(S, W, H, D, minD, p->uniqueness_ratio, disp, raw, mins);
        if (st && st->disp_wta) memcpy(st->disp_wta, disp, npx * 2);
        if (st && st->raw) memcpy(st->raw, raw, npx * 2);
        if (st && st->mins) memcpy(st->mins, mins, npx * 2);
        if (!(p->flags & SVREF_FLAG_NO_LRCHECK)) {
            int thr = p->disp12_max_diff;
            if (p->flags & SVREF_FLAG_LR_OPENCV) thr = thr > 0 ? thr : 1; /* EXT, Appendix A7 */
            svref_lrcheck_min(disp, raw, mins, W, H, D, minD, thr, st ? st->disp2 : NULL);
        }
        if (st && st->disp_lr) memcpy(st->disp_lr, disp, npx * 2);
        if (!(p->flags & SVREF_FLAG_NO_MEDIAN))
            svref_median3(disp, med, W, H, (p->flags & SVREF_FLAG_MEDIAN_REPLICATE) != 0);
        else
            memcpy(med, disp, npx * 2);
        if (st && st->disp_med) memcpy(st->disp_med, med, npx * 2);
        if (p->speckle_window_size > 0)
            svref_filter_speckles(med, W, H, geo_invalid(minD), p->speckle_window_size, 16 * p->speckle_range);
        for (int y = 0; y < H; y++) memcpy((char*)disp16 + y * ds, med + (size_t)y * W, (size_t)W * 2);
    }
    free(l); free(r); free(C); free(S); free(disp); free(raw); free(mins); free(med);
    return rc;
}

/* A11.  EXT cv::ximgproc::getDisparityVis(src, dst, scale) as called at
 * StereoMatcher.cpp:437: dst = saturate_cast<uchar>(src * scale/16), rounded
 * half to even (cvRound). */
void svref_disparity_vis(const int16_t* disp16, uint8_t* vis, size_t n, int scale) {
    for (size_t i = 0; i < n; i++) {
        int v = disp16[i] * scale;
        int q = v >> 4, r = v & 15; /* floor division */
        if (r > 8 || (r == 8 && (q & 1))) q++;
        vis[i] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
    }
}

/* A12.  EXT cv::cvtColor(CV_BGR2GRAY) applied to RGB-ordered bytes
 * (StereoMatcher.cpp:429-430): byte 0 gets the "B" weight. */
void svref_gray_from_rgb(const uint8_t* rgb, uint8_t* gray, size_t n) {
    for (size_t i = 0; i < n; i++)
        gray[i] = (uint8_t)((rgb[3 * i] * 1868 + rgb[3 * i + 1] * 9617 + rgb[3 * i + 2] * 4899 + (1 << 13)) >> 14);
}

/* A13.  EXT cv::remap(src, dst, mapx, mapy, INTER_LINEAR) with CV_32FC1 maps and
 * BORDER_CONSTANT(0) as called at stereoCamera.cpp:573-574: coordinates
 * quantised to 1/32 px (round half even), 15-bit bilinear weights, out-of-image
 * taps read 0. */
void svref_remap_rgb(const uint8_t* src, int W, int H, const float* mapx, const float* mapy, uint8_t* dst) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t o = (size_t)y * W + x;
            float fxm = mapx[o] * 32.0f, fym = mapy[o] * 32.0f;
            /* saturate like cvRound + saturate_cast<short>(>>5) */
            if (!(fxm > -1.0e9f)) fxm = -1.0e9f;
            if (!(fxm < 1.0e9f)) fxm = 1.0e9f;
            if (!(fym > -1.0e9f)) fym = -1.0e9f;
            if (!(fym < 1.0e9f)) fym = 1.0e9f;
            long sx = lrintf(fxm), sy = lrintf(fym);
            long ix = sx >> 5, iy = sy >> 5;
            if (ix < -32768) ix = -32768;
            if (ix > 32767) ix = 32767;
            if (iy < -32768) iy = -32768;
            if (iy > 32767) iy = 32767;
            int fx = (int)(sx & 31), fy = (int)(sy & 31);
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            for (int c = 0; c < 3; c++) {
                int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
                int x0ok = ix >= 0 && ix < W, x1ok = ix + 1 >= 0 && ix + 1 < W;
                int y0ok = iy >= 0 && iy < H, y1ok = iy + 1 >= 0 && iy + 1 < H;
                if (y0ok && x0ok) p00 = src[((size_t)iy * W + ix) * 3 + c];
                if (y0ok && x1ok) p01 = src[((size_t)iy * W + ix + 1) * 3 + c];
                if (y1ok && x0ok) p10 = src[((size_t)(iy + 1) * W + ix) * 3 + c];
                if (y1ok && x1ok) p11 = src[((size_t)(iy + 1) * W + ix + 1) * 3 + c];
                int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
                dst[o * 3 + c] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
}

/* A14.  REPO DispModule.cpp:320-366 (depthFromDisparity).  usign is compared
 * with cols, vsign with rows (:358); coefficients are cast to float first
 * (:333-343).  alias = 1: read the image being overwritten (:361), raster order. */
void svref_depth(const int16_t* disp16, int W, int H, const float* mapper, const double* Q, const double* R,
                 int alias, float* depth) {
    const size_t n = (size_t)W * H;
    float* src = depth;
    for (size_t i = 0; i < n; i++) depth[i] = (float)disp16[i] * (float)(1. / 16.);
    if (!alias) {
        src = (float*)malloc(n * sizeof(float));
        memcpy(src, depth, n * sizeof(float));
    }
    const float r02 = (float)R[2], r12 = (float)R[5], r22 = (float)R[8];
    const float q00 = (float)Q[0], q03 = (float)Q[3], q11 = (float)Q[5], q13 = (float)Q[7];
    const float q23 = (float)Q[11], q32 = (float)Q[14], q33 = (float)Q[15];
    for (int u = 0; u < H; u++)
        for (int v = 0; v < W; v++) {
            size_t o = (size_t)u * W + v;
            float usign = (float)lrintf(mapper[2 * o]);
            float vsign = (float)lrintf(mapper[2 * o + 1]);
            if (usign < 0 || usign >= W || vsign < 0 || vsign >= H)
                depth[o] = 0.0f;
            else {
                float num = r02 * (vsign * q00 + q03) + r12 * (usign * q11 + q13) + r22 * q23;
                float den = src[(size_t)vsign * W + (size_t)usign] * q32 + q33;
                depth[o] = num / den;
            }
        }
    if (!alias) free(src);
}

/* ------------------------------------------------------------------------------------------
 * A15 (EXT).  WLS post-filter: what StereoMatcherNew::filterWLS asks of
 * cv::ximgproc::createDisparityWLSFilterGeneric(false) (StereoMatcher.cpp:127-198, ctor :265):
 * no confidence map, so DisparityWLSFilterImpl::filter reduces to
 *   dst = 16*(min_disp-1) everywhere (min_disp = 0 in the Generic factory), then
 *   dst(ROI) = fastGlobalSmootherFilter(guide(ROI), disp(ROI), lambda, sigma_color)
 * with lambda_attenuation 0.25 and 3 iterations (the defaults of fastGlobalSmootherFilter),
 * the input converted to float, and the result converted back with saturate_cast(cvRound).
 * The fast global smoother (Min et al. 2014) alternates, per iteration, one tridiagonal solve
 * along every row and one along every column of (I + lambda*A) u = f, A the 1-D Laplacian with
 * edge weights exp(-|dI|/sigma) taken from a table indexed by the squared colour distance;
 * lambda is multiplied by the attenuation after each iteration.  OpenCV is not installed, so
 * the operation order below (Thomas algorithm, float) is this oracle's and "parity unpinned".
 * ------------------------------------------------------------------------------------------ */
#define SVREF_WLS_LUT_SIZE (3 * 256 * 256)

void svref_wls_lut(float sigma, float* lut) {
    for (int i = 0; i < SVREF_WLS_LUT_SIZE; i++) lut[i] = -expf(-sqrtf((float)i) / sigma);
}

/* computeROI2, StereoMatcher.cpp:12-29 */
void svref_wls_roi(int W, int H, const svref_params* p, int roi[4]) {
    const int bs2 = p->block_size / 2;
    const int minD = p->min_disparity, maxD = p->min_disparity + p->num_disparities - 1;
    const int xmin = maxD + bs2, xmax = W + minD - bs2, ymin = bs2, ymax = H - bs2;
    roi[0] = xmin; roi[1] = ymin; roi[2] = xmax - xmin; roi[3] = ymax - ymin;
}

/* one Thomas solve over n samples with stride `st`; w[k*st] is the (negative) weight between
 * sample k and k+1 (0 for the last), tmp receives the modified super-diagonal */
static void fgs_solve(float* cur, const float* w, float* tmp, int n, size_t st, float lambda) {
    float coef_cur = lambda * w[0];
    float denom = 1.0f - coef_cur;
    tmp[0] = coef_cur / denom;
    cur[0] = cur[0] / denom;
    for (int k = 1; k < n; k++) {
        const float coef_prev = lambda * w[(size_t)(k - 1) * st];
        coef_cur = lambda * w[(size_t)k * st];
        denom = (1.0f - coef_prev - coef_cur) - coef_prev * tmp[(size_t)(k - 1) * st];
        tmp[(size_t)k * st] = coef_cur / denom;
        cur[(size_t)k * st] = (cur[(size_t)k * st] - coef_prev * cur[(size_t)(k - 1) * st]) / denom;
    }
    for (int k = n - 2; k >= 0; k--) cur[(size_t)k * st] = cur[(size_t)k * st] - tmp[(size_t)k * st] * cur[(size_t)(k + 1) * st];
}

/* fastGlobalSmootherFilter on a float image [h][w] in place; guide is 8UC3 with row stride gs */
int svref_fgs(const uint8_t* guide, size_t gs, float* img, int w, int h, float lambda, float sigma, float att,
              int iters) {
    if (w < 1 || h < 1 || !(sigma > 0) || !(lambda >= 0) || iters < 1) return -1;
    const size_t n = (size_t)w * h;
    float* lut = (float*)malloc(sizeof(float) * SVREF_WLS_LUT_SIZE);
    float* ch = (float*)malloc(sizeof(float) * n);
    float* cv = (float*)malloc(sizeof(float) * n);
    float* tmp = (float*)malloc(sizeof(float) * n);
    if (!lut || !ch || !cv || !tmp) { free(lut); free(ch); free(cv); free(tmp); return -2; }
    svref_wls_lut(sigma, lut);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const uint8_t* p = guide + (size_t)i * gs + 3 * (size_t)j;
            float wh = 0.0f, wv = 0.0f;
            if (j < w - 1) {
                const uint8_t* q = p + 3;
                wh = lut[(p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2])];
            }
            if (i < h - 1) {
                const uint8_t* q = p + gs;
                wv = lut[(p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2])];
            }
            ch[(size_t)i * w + j] = wh;
            cv[(size_t)i * w + j] = wv;
        }
    float lam = lambda;
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < h; i++) fgs_solve(img + (size_t)i * w, ch + (size_t)i * w, tmp + (size_t)i * w, w, 1, lam);
        for (int j = 0; j < w; j++) fgs_solve(img + j, cv + j, tmp + j, h, (size_t)w, lam);
        lam *= att;
    }
    free(lut); free(ch); free(cv); free(tmp);
    return 0;
}

/* filterWLS on the 16-bit map (StereoMatcher.cpp:160) and, when vis != NULL, on the 8-bit
 * visualisation (:159).  guide = rectified left image, 8UC3 dense.  roi = {x, y, w, h}.     */
int svref_wls_filter(const int16_t* disp16, const uint8_t* vis, const uint8_t* guide_rgb, int W, int H, const int roi[4],
                     float lambda, float sigma, int16_t* out16, uint8_t* out8) {
    const int rx = roi[0], ry = roi[1], rw = roi[2], rh = roi[3];
    if (rx < 0 || ry < 0 || rw < 1 || rh < 1 || rx + rw > W || ry + rh > H) return -1;
    float* buf = (float*)malloc(sizeof(float) * (size_t)rw * rh);
    if (!buf) return -2;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && (!vis || !out8)) break;
        for (int i = 0; i < rh; i++)
            for (int j = 0; j < rw; j++) {
                const size_t o = (size_t)(ry + i) * W + rx + j;
                buf[(size_t)i * rw + j] = pass == 0 ? (float)disp16[o] : (float)vis[o];
            }
        int rc = svref_fgs(guide_rgb + ((size_t)ry * W + rx) * 3, (size_t)W * 3, buf, rw, rh, lambda, sigma, 0.25f, 3);
        if (rc) { free(buf); return rc; }
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t o = (size_t)y * W + x;
                const int in = x >= rx && x < rx + rw && y >= ry && y < ry + rh;
                const float f = in ? buf[(size_t)(y - ry) * rw + (x - rx)] : -16.0f;
                long r = lrintf(f);  /* cvRound: round half to even */
                if (pass == 0) out16[o] = (int16_t)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
                else out8[o] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
    }
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Rectification-map generation (EXT).  StereoCamera::rectifyImages calls
 * cv::initUndistortRectifyMap(K, Dist, Rrect, Prect, size, CV_32FC1, mapx, mapy)
 * (lib/src/stereoCamera.cpp:569-570) and StereoCamera::updateMappings calls
 * cv::undistortPoints(grid, grid, K, Dist, Rrect, Prect) (:2462-2463) on every frame whose
 * kinematics changed.  Restated from OpenCV's published algorithm, double precision:
 *   dist = {k1, k2, p1, p2, k3, k4, k5, k6} (the rational model; the module fills k1 k2 p1 p2,
 *   DispModule.cpp:927-987); thin-prism and tilt terms are zero in this module and not modelled.
 * initUndistortRectifyMap: iR = inv(P[:3,:3] * R); for each destination pixel (u, v) the ray
 *   iR*(u,v,1) is normalised, distorted and projected with K.  OpenCV walks a row with running
 *   sums (_x += iR[0] ...); the same order is kept here.
 * undistortPoints: pixel -> normalised, 5 fixed-point iterations of the inverse distortion,
 *   then (P[:3,:3]*R) applied and dehomogenised.
 * ------------------------------------------------------------------------------------------ */
static int inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    if (det == 0.0) return -1;
    const double id = 1.0 / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id;  o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
    return 0;
}
static void mul3(const double* a, const double* b, double* o) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}

/* K 3x3, dist[8], R 3x3, P 3x4 (row-major doubles) -> mapx, mapy [H][W] float */
int svref_init_rectify_map(const double* K, const double* dist, const double* R, const double* P, int W, int H,
                           float* mapx, float* mapy) {
    const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double PR[9], ir[9];
    mul3(P3, R, PR);
    if (inv3(PR, ir)) return -1;
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4], k4 = dist[5], k5 = dist[6], k6 = dist[7];
    for (int i = 0; i < H; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double w = 1. / _w, x = _x * w, y = _y * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            mapx[(size_t)i * W + j] = (float)(fx * xd + u0);
            mapy[(size_t)i * W + j] = (float)(fy * yd + v0);
        }
    }
    return 0;
}

/* the inverse mapper of updateMappings: mapper[y][x] = (x', y') rectified coordinates of the
 * original pixel (x, y); [H][W][2] float */
int svref_undistort_grid(const double* K, const double* dist, const double* R, const double* P, int W, int H,
                         float* mapper) {
    const double P3[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    double RR[9];
    mul3(P3, R, RR);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    if (fx == 0.0 || fy == 0.0) return -1;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double* k = dist; /* k1 k2 p1 p2 k3 k4 k5 k6 */
    for (int yi = 0; yi < H; yi++)
        for (int xi = 0; xi < W; xi++) {
            double x = ((double)(float)xi - cx) * ifx, y = ((double)(float)yi - cy) * ify;
            const double x0 = x, y0 = y;
            for (int it = 0; it < 5; it++) {
                const double r2 = x * x + y * y;
                const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
                const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
                x = (x0 - dX) * icdist;
                y = (y0 - dY) * icdist;
            }
            const double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5];
            const double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
            mapper[2 * ((size_t)yi * W + xi)] = (float)(xx * ww);
            mapper[2 * ((size_t)yi * W + xi) + 1] = (float)(yy * ww);
        }
    return 0;
}
