"""Host-to-host throughput of the in-process dispatcher (sv_pool_*) on ONE GPU: a pool of one context against a pool
of two contexts on device 0, against sv_submit_gray_batch_host / sv_wait_batch_host called directly.  C2, pinned
buffers.  What it measures is the dispatcher's own overhead (host threads, chunk dealing), not scaling: both
contexts share one device.  perf_pool.py [pairs per call] [batch per chunk]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stereo-vision_amd"))
import numpy as np
import svhip
from svhip import synth
W, H, D = 640, 480, 128
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pairs = [synth.stereo_pair(202 + i, W, H, D) for i in range(8)]
L = svhip.PinnedArray((n, H, W), np.uint8); R = svhip.PinnedArray((n, H, W), np.uint8); O = svhip.PinnedArray((n, H, W), np.int16)
for i in range(n):
    L.array[i], R.array[i] = pairs[i % 8][0], pairs[i % 8][1]
p = svhip.default_params(D, 8)
lefts, rights, outs = [L.array[i] for i in range(n)], [R.array[i] for i in range(n)], [O.array[i] for i in range(n)]
print("C2, %d pairs per call, chunks of %d pairs, pinned host buffers, host to host" % (n, batch))
# direct: one context, two chunks in flight
m = svhip.Matcher(W, H, p); m.set_batch(batch)
def direct():
    t = []
    for f0 in range(0, n, batch):
        if len(t) == 2: m.wait_batch(t.pop(0))
        t.append(m.submit_gray_batch(lefts[f0:f0 + batch], rights[f0:f0 + batch], outs[f0:f0 + batch]))
    for x in t: m.wait_batch(x)
direct(); t0 = time.perf_counter(); direct(); direct(); dt = (time.perf_counter() - t0) / 2
ref = O.array.copy()
print("  sv_submit_gray_batch_host / sv_wait_batch_host, one context : %7.1f frames/s" % (n / dt), flush=True)
m.close()
for devs in ([0], [0, 0]):
    pool = svhip.Pool(devs, W, H, p, batch)
    O.array[:] = -32768
    pool.compute_gray(lefts, rights, outs)
    assert np.array_equal(O.array, ref), "pool result differs"
    t0 = time.perf_counter(); pool.compute_gray(lefts, rights, outs); pool.compute_gray(lefts, rights, outs); dt = (time.perf_counter() - t0) / 2
    print("  sv_pool_compute_gray_host, %d context(s) on device 0          : %7.1f frames/s" % (len(devs), n / dt), flush=True)
    pool.close()
