// EXPERIMENT (round 2, not part of libsv_hip): a barrier-free form of the cost kernel, bit-exact (all GPU tests passed
// with it), but slower than k_cost2 on MI355X: 33.9 ms per 1024 C2 frames against 27.1 ms; 27.4 ms with its stores switched
// off, 69 ms with direct 16-byte-per-lane stores (no LDS transpose).  Both forms are bound by the VALU work of the
// Birchfield-Tomasi arithmetic (23 instructions per pair of disparities and pixel), not by LDS, barriers or HBM.
// Kept for the record; see DESIGN.md section 4.
// ---------------------------------------------------------------------------------------------
// Stage 2, second form: the same C as k_cost2, no LDS and no barrier.
// k_cost2 gives a tile column to 16 lanes (one disparity group each) and lets the 2r+1-tap horizontal sum
// travel through LDS, one workgroup barrier per image row.  Here a LANE is a tile column (64 columns per
// wave, 64 - 2r of them outputs) and a WAVE is a disparity group (2N disparities; the 16 waves of a
// workgroup cover all D of the same 64 columns).  The box sum runs vertical first: every lane keeps the BT
// costs of its column for the last 2r+1 rows in registers (a ring with compile-time slots: the row loop is
// unrolled by 2r+1) and the vertical running sum `acc`; the horizontal sum of the 2r+1 neighbouring
// columns is then 2r wave-wide DPP shifts folded into adds (v_add_u32_dpp wave_shr:1: lane i ends with
// columns i-2r .. i, i.e. the window centred on tile column i - r).  Waves never wait for each other.
// A lane stores its 16 bytes (its disparity group of its pixel); the 16 waves of the workgroup fill the
// pixel's 32 * N bytes between them, which the XCD's L2 merges before it writes the lines back.
// Same rows / columns / quirks as k_cost2 (see there); R = radius.
// ---------------------------------------------------------------------------------------------
template <int TAPS>
__device__ __forceinline__ uint32_t wave_window_sum(uint32_t x) {
    uint32_t t = x;
#pragma unroll
    for (int i = 1; i < TAPS; i++) t = x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    return t;
}

#ifndef SV_COST3_EXP
#define SV_COST3_EXP 0   // diagnostic builds: 1 = only one lane per wave stores, 3 = aligned (wrong) right-statistics loads
#endif
template <int N>
__host__ __device__ constexpr int cost3_tile_pitch() { return 16 * N + 4; }   // dwords per output column of the LDS tile (+4: conflict-free 16-byte writes)
template <int N, int R>
__host__ __device__ constexpr size_t cost3_lds_bytes() { return (size_t)2 * (64 - 2 * R) * cost3_tile_pitch<N>() * 4; }

template <int N, int R, int CN>
__global__ __launch_bounds__(1024) void k_cost3(const uint2* __restrict__ stat_l, const uint16_t* __restrict__ rp,
                                                 uint32_t* __restrict__ C, int W, int H, int WP, int yband, uint32_t P2P2,
                                                 int tail_p2, int ntx, int nbands, int nframes) {
    constexpr int cn = CN, D = 32 * N, D2 = 16 * N, RING = 2 * R + 1, TX = 64 - 2 * R, PITCH = cost3_tile_pitch<N>();
    extern __shared__ __align__(16) uint32_t c3_lds[];   // [2][TX][PITCH]: a row of C as the window sums leave the waves
    const int W1 = W - D, tid = threadIdx.x, j = tid & 63, sl = tid >> 6;   // tile column, disparity group
    const int per_frame = ntx * nbands;
    const int wg = blockIdx.x;
    const bool by_xcd = nframes > 0;   // see k_cost2: id % 8 picks the frame, a frame's tiles share an XCD's L2
    const int fr = by_xcd ? (wg / 8 / per_frame) * 8 + (wg & 7) : wg / per_frame;
    const int tb = by_xcd ? (wg / 8) % per_frame : wg % per_frame;
    if (fr >= (by_xcd ? nframes : -nframes)) return;
    const int xl0 = (tb % ntx) * TX;
    const int y0 = (tb / ntx) * yband, y1 = min(y0 + yband, H);
    stat_l += (size_t)fr * cn * W * H;
    const uint32_t* rp32 = reinterpret_cast<const uint32_t*>(rp + (size_t)fr * cn * 6 * H * WP);
    C += (size_t)fr * H * W1 * D2;
    const int xj = min(max(xl0 - R + j, 0), W1 - 1) + D;   // this lane's image column (clamped into the computable band)
    int e = (W - 1 - xj) + 2 * N * sl;                     // reversed index of x - d for the group's first disparity
    if (SV_COST3_EXP == 3) e &= ~7;
    const int cpy = e & 1;
    const size_t plane = (size_t)H * WP / 2;
    const uint32_t* rbase = rp32 + (size_t)cpy * plane + (size_t)((e - cpy) >> 1);
    // compute role: this lane's window sum belongs to output column j - 2R of the tile
    const bool wr_ok = j >= 2 * R;
    uint32_t* const tile_w = c3_lds + (size_t)max(j - 2 * R, 0) * PITCH + sl * N;
    // store role: thread (column tid / 16, disparity group tid % 16) moves 16 bytes of the finished row to C
    const int scol = tid >> 4, sq = tid & 15;
    const bool st_ok = scol < TX && xl0 + scol < W1;
    const uint32_t* const tile_r = c3_lds + (size_t)min(scol, TX - 1) * PITCH + sq * N;
    uint32_t* const Cout = C + (size_t)min(xl0 + scol, W1 - 1) * D2 + sq * N;
    const size_t rowN = (size_t)W1 * D2;

    uint2 lrec;
    uint32_t rq[3][N];
    auto prefetch = [&](int ry, int ch) {
        ry = min(ry, H - 1);
        lrec = stat_l[(size_t)ch * W * H + (size_t)ry * W + xj];
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++)
            load_regs<N>(rbase + (size_t)(6 * ch + 2 * s3) * plane + (size_t)ry * (WP / 2), rq[s3]);
    };
    // BT costs of image row ry for this lane's column and disparity group (operands prefetched), summed over the colour planes
    auto bt_row = [&](int ry, uint32_t (&pv)[N]) {
#pragma unroll
        for (int k = 0; k < N; k++) pv[k] = 0;
#pragma unroll 1
        for (int ch = 0; ch < cn; ch++) {
            const uint32_t ug = (lrec.x & 0xFFu) * 0x10001u, ug0 = ((lrec.x >> 8) & 0xFFu) * 0x10001u;
            const uint32_t ug1 = ((lrec.x >> 16) & 0xFFu) * 0x10001u, ur = (lrec.x >> 24) * 0x10001u;
            const uint32_t ur0 = (lrec.y & 0xFFu) * 0x10001u, ur1 = ((lrec.y >> 8) & 0xFFu) * 0x10001u;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint32_t vg = rq[0][k] & 0x00FF00FFu, vg0 = rq[1][k] & 0x00FF00FFu, vg1 = rq[2][k] & 0x00FF00FFu;
                const uint32_t vr = __builtin_amdgcn_perm(0u, rq[0][k], 0x0C030C01u), vr0 = __builtin_amdgcn_perm(0u, rq[1][k], 0x0C030C01u);
                const uint32_t vr1 = __builtin_amdgcn_perm(0u, rq[2][k], 0x0C030C01u);
                pv[k] += bt_pk2(ug, ug0, ug1, vg, vg0, vg1) + __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, bt_pk2(ur, ur0, ur1, vr, vr0, vr1)) >> 2);
            }
            if (ch + 1 < cn) prefetch(ry, ch + 1);
            else prefetch(ry + 1, 0);
        }
    };

    const int ylast = H - 1 - R;
    const int ys = min(y0, ylast);
    const int r_begin = max(ys - R, 0), r_init_end = ys + R;
    const int r_end = max(min(y1 - 1, ylast), ys) + R;
    uint32_t hist[RING][N], acc[N], pend[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = pend[k] = 0;
    int pend_y = -1;   // row whose 16 bytes wait in `pend` (read back from the tile after the barrier; stored a row later,
                       // behind the next row's arithmetic, so that the LDS round trip is never waited for)
    auto flush = [&]() {
        if (pend_y >= 0 && st_ok && (SV_COST3_EXP != 1 || sq == 0)) store_regs<N>(Cout + (size_t)pend_y * rowN, pend);
    };
    // the window sum of the vertical sums, + P2, to C[y = ry - R] once the window is complete
    auto emit = [&](int ry) {
        const int y = ry - R;
        flush();
        pend_y = -1;
        if (ry >= r_init_end) {   // (a band that lies in the bottom rows only needs this row for them: y < y0 there)
            uint32_t out[N];
#pragma unroll
            for (int k = 0; k < N; k++) out[k] = wave_window_sum<RING>(acc[k]) + P2P2;
            const int par = (ry & 1) * (TX * PITCH);
            if (wr_ok) store_regs<N>(tile_w + par, out);
            lds_barrier();
            load_regs<N>(tile_r + par, pend);
            pend_y = y >= y0 ? y : -1;
        }
    };
    prefetch(r_begin, 0);
    int ry = r_begin;
    // the first 2r+1 rows: the window fills (row 0 stands for every row above the image), then rows that
    // leave it at the top are replicas of row 0 (slot 0)
#pragma unroll
    for (int i = 0; i < RING; i++) {
        if (ry <= r_end) {
            uint32_t pv[N];
            bt_row(ry, pv);
            if (ry <= r_init_end) {
                const uint32_t mult = ry == 0 ? (uint32_t)(R - ys + 1) : 1u;
#pragma unroll
                for (int k = 0; k < N; k++) acc[k] += pv[k] * mult;
            } else {
#pragma unroll
                for (int k = 0; k < N; k++) acc[k] = acc[k] + pv[k] - hist[0][k];
            }
#pragma unroll
            for (int k = 0; k < N; k++) hist[i][k] = pv[k];
            emit(ry);
            ry++;
        }
    }
    // steady state: the row that leaves the window sits in the slot the new row takes
    while (ry <= r_end) {
#pragma unroll
        for (int u = 0; u < RING; u++) {
            if (ry <= r_end) {
                uint32_t pv[N];
                bt_row(ry, pv);
#pragma unroll
                for (int k = 0; k < N; k++) {
                    acc[k] = acc[k] + pv[k] - hist[u][k];
                    hist[u][k] = pv[k];
                }
                emit(ry);
                ry++;
            }
        }
    }
    flush();
    // the bottom `radius` rows repeat the last computed row (costs.cu:279-282: still in `pend`), or hold P2
    for (int y = max(y0, ylast + 1); y < y1; y++) {
        uint32_t o[N];
#pragma unroll
        for (int k = 0; k < N; k++) o[k] = tail_p2 ? P2P2 : pend[k];
        if (st_ok) store_regs<N>(Cout + (size_t)y * rowN, o);
    }
}

