"""Host-to-host throughput of sv_submit_gray_batch_host / sv_wait_batch_host (C2), pinned and pageable buffers."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stereo-vision_amd'))
import numpy as np
import svhip
from svhip import synth
W, H, D = 640, 480, 128
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(8)]
m = svhip.Matcher(W, H, svhip.default_params(D, 8))
for nb in (32, 128, 256, 512):
    m.set_batch(nb)
    for pinned in (True, False):
        bufs = []
        for s in range(2):   # two sets of buffers: two batches in flight
            if pinned:
                l, r, o = svhip.PinnedArray((nb, H, W), np.uint8), svhip.PinnedArray((nb, H, W), np.uint8), svhip.PinnedArray((nb, H, W), np.int16)
                la, ra, oa = l.array, r.array, o.array
            else:
                l = r = o = None
                la, ra, oa = np.empty((nb, H, W), np.uint8), np.empty((nb, H, W), np.uint8), np.empty((nb, H, W), np.int16)
            for i in range(nb):
                la[i], ra[i] = pairs[i % 8][:2]
            bufs.append((la, ra, oa, (l, r, o)))
        def submit(s):
            la, ra, oa, _ = bufs[s]
            return m.submit_gray_batch([la[i] for i in range(nb)], [ra[i] for i in range(nb)], [oa[i] for i in range(nb)])
        t0 = submit(0); m.wait_batch(t0)          # warm-up (allocations)
        t0 = submit(1); m.wait_batch(t0)
        reps = max(4, 2048 // nb)
        t = time.perf_counter()
        tick = submit(0)
        for k in range(1, reps):
            nxt = submit(k & 1)
            m.wait_batch(tick)
            tick = nxt
        m.wait_batch(tick)
        dt = time.perf_counter() - t
        ok = (bufs[0][2][:, :, D:] >= 0).mean()
        print("batch %4d %-8s: %8.1f frames/s host to host (%d batches, valid %.3f)" % (nb, "pinned" if pinned else "pageable", reps * nb / dt, reps, ok), flush=True)
        for b in bufs:
            for a in b[3]:
                if a is not None: a.free()
m.close()
