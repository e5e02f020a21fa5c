"""naive_sgm.py — second, independent restatement of the SGBM arithmetic (numpy).

TEST INFRASTRUCTURE ONLY (see oracle/sv_ref.h).  Parity unpinned: the reference
holds no golden vectors for this path, so the C oracle (sv_ref.c) is pinned by
differential testing against this file, which is written separately and in a
different style on purpose: direct window sums instead of running sums, whole
planes instead of scanlines, pure per-pixel loops for the recurrence.

Follows SURVEY.md Appendix A; reference anchors (relative to
/root/reference/modules/DisparityModule/sgbm_cuda/):
  prefilter   costs.cu:116-166, 286-292      BT cost   costs.cu:170-201
  box sums    costs.cu:203-284               paths     aggregation.cu:60-625
  WTA         aggregation.cu:629-689         LR check  aggregation.cu:691-735
  median      aggregation.cu:737-768

Only small images are practical (O(W*H*D) Python-level work per path).
"""
import numpy as np

INVALID = -16
DIRS = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]

FLAG_C_COL0_P2 = 0x01
FLAG_C_TAIL_P2 = 0x02
FLAG_LR_OPENCV = 0x04
FLAG_MEDIAN_REPLICATE = 0x08
FLAG_NO_LRCHECK = 0x10
FLAG_NO_MEDIAN = 0x20


def prefilter(img, cap):
    """A1: clipped Sobel-x; edge columns of grad AND raw forced to ftzero."""
    I = img.astype(np.int32)
    H, W = I.shape
    up = np.vstack([I[:1], I[:-1]])
    dn = np.vstack([I[1:], I[-1:]])
    g = np.full((H, W), cap, np.int32)
    v = 2 * (I[:, 2:] - I[:, :-2]) + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2])
    g[:, 1:-1] = np.clip(v, -cap, cap) + cap
    raw = I.copy()
    raw[:, 0] = cap
    raw[:, -1] = cap
    return g.astype(np.uint8), raw.astype(np.uint8)


def _halfpel_minmax(row):
    """per-pixel (v, v0, v1) of the BT interval along a row (int32)."""
    v = row.astype(np.int32)
    left = np.concatenate([v[:1], v[:-1]])
    right = np.concatenate([v[1:], v[-1:]])
    vl = (v + left) // 2
    vr = (v + right) // 2
    vl[0] = v[0]
    vr[-1] = v[-1]
    return v, np.minimum(np.minimum(vl, vr), v), np.maximum(np.maximum(vl, vr), v)


def columns(W, D, minD):
    """computable columns [col0, colN) for a disparity range [minD, minD + D) (EXT: cv::StereoSGBM minDisparity)."""
    return max(minD + D, 0), W + min(minD, 0)


def bt_cost(L, R, D, shift, minD=0):
    """A2 for one channel -> [H][W1][D] int32."""
    H, W = L.shape
    col0, colN = columns(W, D, minD)
    W1 = colN - col0
    out = np.zeros((H, W1, D), np.int32)
    for y in range(H):
        u, u0, u1 = _halfpel_minmax(L[y])
        v, v0, v1 = _halfpel_minmax(R[y])
        for d in range(D):
            xs = np.arange(col0, colN)
            xr = xs - (d + minD)
            c0 = np.maximum(np.maximum(0, u[xs] - v1[xr]), v0[xr] - u[xs])
            c1 = np.maximum(np.maximum(0, v[xr] - u1[xs]), u0[xs] - v[xr])
            out[y, :, d] = np.minimum(c0, c1) >> shift
    return out


def cost_volume(left, right, D, BS, P2, cap, flags=0, minD=0):
    """A2-A4 as direct clamped-window sums."""
    gl, rl = prefilter(left, cap)
    gr, rr = prefilter(right, cap)
    pix = bt_cost(gl, gr, D, 0, minD) + bt_cost(rl, rr, D, 2, minD)
    H, W1, _ = pix.shape
    r = BS // 2
    hs = np.zeros_like(pix)
    for k in range(-r, r + 1):
        idx = np.clip(np.arange(W1) + k, 0, W1 - 1)
        hs += pix[:, idx, :]
    C = np.zeros_like(pix)
    ylast = H - 1 - r
    for y in range(H):
        yy = min(y, ylast)  # bottom r rows copy the last computed row
        acc = np.full((W1, D), P2, np.int32)
        for k in range(-r, r + 1):
            acc += hs[max(yy + k, 0)]  # yy + k <= H-1 by construction
        C[y] = acc
    C[:, 0, :] = C[0, 0, :]  # x = D column quirk: never updated after row 0
    if flags & FLAG_C_TAIL_P2:
        C[max(H - r, 1):, :, :] = P2
    if flags & FLAG_C_COL0_P2:
        C[1:, 0, :] = P2
    return C.astype(np.int16), dict(grad_l=gl, grad_r=gr, raw_l=rl, raw_r=rr, pix=pix.astype(np.int16),
                                    hsum=hs.astype(np.int16))


def aggregate_path(C, P1, P2, direction):
    """A5 for one direction -> L [H][W1][D] int32 (per-pixel loop)."""
    dx, dy = DIRS[direction]
    H, W1, D = C.shape
    L = np.zeros((H, W1, D), np.int32)
    ys = range(H) if dy >= 0 else range(H - 1, -1, -1)
    xs = list(range(W1)) if dx >= 0 else list(range(W1 - 1, -1, -1))
    BIG = 32767
    for y in ys:
        for x in xs:
            px, py = x - dx, y - dy
            if 0 <= px < W1 and 0 <= py < H:
                lp = L[py, px]
                delta = int(lp.min()) + P2
            else:
                lp = np.zeros(D, np.int32)
                delta = P2
            lm = np.concatenate([[BIG], lp[:-1]]) + P1
            lq = np.concatenate([lp[1:], [BIG]]) + P1
            m = np.minimum(np.minimum(lp, lm), np.minimum(lq, delta))
            L[y, x] = C[y, x].astype(np.int32) + m - delta
    return L


def aggregate(C, P1, P2, path_mask):
    S = np.zeros(C.shape, np.int32)
    Ls = {}
    for k in range(8):
        if path_mask & (1 << k):
            Ls[k] = aggregate_path(C, P1, P2, k)
            S = np.minimum(S + Ls[k], 32767)
    return S.astype(np.int16), Ls


def wta(S, W, uniq, minD=0):
    """A6 -> disp, raw, mins (full width).  raw is the cost index; disp carries minD (EXT)."""
    H, W1, D = S.shape
    col0 = columns(W, D, minD)[0]
    disp = np.full((H, W), (minD - 1) * 16, np.int16)
    raw = np.full((H, W), -1, np.int16)
    mins = np.full((H, W), 32767, np.int16)
    ds = np.arange(D)
    for y in range(H):
        for xl in range(W1):
            s = S[y, xl].astype(np.int64)
            x = xl + col0
            minS = int(s.min())
            if minS >= 32767:
                continue
            best = int(np.argmax(s == minS))  # first minimum
            mins[y, x] = minS
            if np.any((s * (100 - uniq) < minS * 100) & (np.abs(ds - best) > 1)):
                continue
            raw[y, x] = best
            if 0 < best < D - 1:
                den = max(int(s[best - 1] + s[best + 1] - 2 * s[best]), 1)
                num = int(s[best - 1] - s[best + 1]) * 16 + den
                q = abs(num) // (2 * den)  # C division truncates toward zero
                disp[y, x] = (best + minD) * 16 + (q if num >= 0 else -q)
            else:
                disp[y, x] = (best + minD) * 16
    return disp, raw, mins


def lrcheck(disp, raw, mins, D, thresh, minD=0):
    """A7 (with min_disparity as cv::StereoSGBM has it: disp2 holds true disparities; an unwritten entry keeps the scaled
    marker (minD - 1) * 16, which the test "disp2 >= minD" takes for a disparity once minD >= 2)."""
    H, W = disp.shape
    col0, colN = columns(W, D, minD)
    inv = (minD - 1) * 16
    out = disp.copy()
    for y in range(H):
        d2 = np.full(W, (minD - 1) * 16, np.int32)
        c2 = np.full(W, 32767, np.int32)
        for x in range(colN - 1, col0 - 1, -1):
            d = int(raw[y, x])
            if d < 0:
                continue
            x2 = x - d - minD
            if c2[x2] > mins[y, x]:
                c2[x2] = mins[y, x]
                d2[x2] = d + minD
        for x in range(col0, colN):
            d1 = int(disp[y, x])
            if d1 == inv:
                continue
            lo, hi = d1 >> 4, (d1 + 15) >> 4
            xa, xb = x - lo, x - hi
            if (0 <= xa < W and d2[xa] >= minD and abs(d2[xa] - lo) > thresh and 0 <= xb < W and d2[xb] >= minD
                    and abs(d2[xb] - hi) > thresh):
                out[y, x] = inv
    return out


def median3(img, replicate):
    """A8."""
    H, W = img.shape
    p = np.pad(img, 1, mode="edge")
    stack = np.stack([p[i:i + H, j:j + W] for i in range(3) for j in range(3)])
    med = np.sort(stack, axis=0)[4]
    if not replicate:
        med[0, :], med[-1, :], med[:, 0], med[:, -1] = img[0, :], img[-1, :], img[:, 0], img[:, -1]
    return med.astype(np.int16)


def filter_speckles(img, new_val, max_size, max_diff):
    """A10 via union-find (a different algorithm from the oracle's flood fill)."""
    H, W = img.shape
    parent = list(range(H * W))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    v = img.astype(np.int32)
    for y in range(H):
        for x in range(W):
            if v[y, x] == new_val:
                continue
            for yy, xx in ((y, x + 1), (y + 1, x)):
                if yy < H and xx < W and v[yy, xx] != new_val and abs(v[yy, xx] - v[y, x]) <= max_diff:
                    a, b = find(y * W + x), find(yy * W + xx)
                    if a != b:
                        parent[a] = b
    roots = np.array([find(i) for i in range(H * W)])
    counts = np.bincount(roots, minlength=H * W)
    out = img.copy().reshape(-1)
    kill = (counts[roots] <= max_size) & (out != new_val)
    out[kill] = new_val
    return out.reshape(H, W)


def sgbm(left, right, D, BS=7, P1=None, P2=None, cap=63, uniq=15, disp12=0, path_mask=0x1F, flags=0,
         speckle_window=0, speckle_range=0, minD=0):
    P1 = 8 * BS * BS if P1 is None else P1
    P2 = 32 * BS * BS if P2 is None else P2
    H, W = left.shape
    C, st = cost_volume(left, right, D, BS, P2, cap, flags, minD)
    S, Ls = aggregate(C, P1, P2, path_mask)
    disp, raw, mins = wta(S, W, uniq, minD)
    st.update(C=C, S=S, L=Ls, disp_wta=disp, raw=raw, mins=mins)
    if not flags & FLAG_NO_LRCHECK:
        thr = disp12
        if flags & FLAG_LR_OPENCV:
            thr = thr if thr > 0 else 1
        disp = lrcheck(disp, raw, mins, D, thr, minD)
    st["disp_lr"] = disp
    if not flags & FLAG_NO_MEDIAN:
        disp = median3(disp, bool(flags & FLAG_MEDIAN_REPLICATE))
    st["disp_med"] = disp
    if speckle_window > 0:
        disp = filter_speckles(disp, (minD - 1) * 16, speckle_window, 16 * speckle_range)
    return disp, st
