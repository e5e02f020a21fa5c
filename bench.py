#!/usr/bin/env python3
"""bench.py — disparity frames/s per GPU at 640x480, D=128, 8 paths (+ LR check), % of HBM roofline.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1: plain process.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL is used
  only for the timing barrier / max-over-ranks; stereo pairs are independent, so frames are
  sharded across ranks with NO data-path collective -> "scaling": "weak").

A "step" is one pass of the hot path (prefilter -> cost volume -> 8-path SGM -> WTA -> LR check ->
median; sv_compute_gray_batch_dev) over one batch of synthetic stereo pairs that are already
resident in HBM.  value = frames of all ranks / max-over-ranks wall time of exactly K steps.

roofline: the dominant kernel is the aggregation kernel: svk::k_sweep (four SGM paths of a sweep over
the whole batch per launch; large batches, D <= 128) or svk::k_agg (one path per launch).  Its
average launch duration is measured live with HIP events recorded by libsv_hip on the stream the
kernels run on, over the K timed steps; achieved = algorithmic bytes per launch (3 * V * batch per
path, V = (W-D)*H*D*2, SURVEY.md 8(d): 12 * V * batch for a four-path sweep) / that duration.  The
fused sweep moves far fewer bytes than that (traffic = the PMC figure), so its algorithmic rate can
exceed the HBM peak; hbm_actual_gbs = traffic / duration is what the memory system really carries.

cpu_baseline: the CPU oracle (oracle/libsv_ref.so, a restatement = "port"; OpenCV is not
installed) timed single-threaded on rank 0 on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))

CONFIGS = {
    # name: (W, H, D, paths)
    "c1": (320, 240, 64, 8),
    "c2": (640, 480, 128, 8),   # the configuration BASELINE.json's metric is quoted on
    "c4": (1920, 1080, 256, 8),
    # BASELINE.json configs[2]: rectify remap + SGM + WLS + depth, 640x480, device-resident RGB pairs of
    # one camera pair (not the headline metric; run with --config c3)
    "c3": (640, 480, 128, 8),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(W, H, D, paths, left, right, budget_s=10.0, threads=16):
    """Times the oracle on the host cores: frames/s on a bounded sample, one thread (how cv::StereoSGBM runs
    MODE_SGBM / MODE_HH) and frame-parallel on `threads` threads (ctypes releases the GIL inside the C oracle)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import svref  # oracle: checker / CPU comparator only, never on the product path
    from concurrent.futures import ThreadPoolExecutor
    p = svref.default_params(D, paths)
    svref.sgbm(left[0], right[0], p)  # page the library in
    n, t0 = 0, time.perf_counter()
    while True:
        svref.sgbm(left[n % len(left)], right[n % len(right)], p)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    per_frame = el / n
    out = {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of %dx%d D=%d %d-path SGBM in %.1f s, oracle/libsv_ref.so (restated SGBM; OpenCV not "
                     "installed), single thread" % (n, W, H, D, paths, el)}
    try:
        threads = max(1, min(threads, len(os.sched_getaffinity(0))))
        each = max(1, int(round(budget_s / per_frame)))   # frames per thread: about budget_s of wall time

        def work(k):
            for i in range(each):
                svref.sgbm(left[(k + i) % len(left)], right[(k + i) % len(right)], p)
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, range(threads)))
        el2 = time.perf_counter() - t1
        out["frame_parallel"] = {"value": round(threads * each / el2, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                                 "sample": "%d frames (%d threads x %d) in %.1f s" % (threads * each, threads, each, el2)}
    except Exception as e:  # the single-thread figure stands on its own
        out["frame_parallel_error"] = str(e)[:200]
    # the reference's own LIBELAS CPU matcher, if oracle/_ref was built in the dev container
    try:
        import ctypes
        import numpy as np
        so = os.path.join(ROOT, "oracle", "_ref", "libelas_ref.so")
        if os.path.exists(so):
            L = ctypes.CDLL(so)
            L.elasref_process.restype = ctypes.c_int
            d1 = np.empty((H, W), np.float32)
            d2 = np.empty((H, W), np.float32)
            args = [left[0].ctypes.data_as(ctypes.c_void_p), right[0].ctypes.data_as(ctypes.c_void_p),
                    d1.ctypes.data_as(ctypes.c_void_p), d2.ctypes.data_as(ctypes.c_void_p), W, H, D - 1, 40]
            L.elasref_process(*args)
            m, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < 3.0:
                L.elasref_process(*args)
                m += 1
            out["libelas"] = {"value": round(m / (time.perf_counter() - t1), 2), "unit": "frames/s",
                              "kind": "reference", "cores": int(L.elasref_threads()),
                              "sample": "%d frames, Elas::process ROBOTICS preset (elas_omp.cpp)" % m}
    except Exception as e:  # comparator is optional
        out["libelas_error"] = str(e)[:200]
    return out


def pmc_traffic(config, batch, kernel):
    """HBM bytes per aggregation launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this workload, gfx950 FETCH_SIZE x2 correction applied);
    None when no pass was collected for this config/batch/kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f).get("%s_b%d%s" % (config, batch, "_sweep" if kernel == "svk::k_sweep" else ""))
        return int(t["traffic_bytes_per_launch"]) if t else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="stereo pairs per step per GPU (C2: 1024 pairs = 140 GB of the 288 GB of HBM; every workgroup pipeline then streams 32 frames and the fill/drain of the band pipeline is 5 %% of a launch instead of 10 %% at 512)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        # RCCL carries only the timing barrier and the max-over-ranks; if it cannot come up on this node the
        # same two operations go over gloo (host tensors) rather than losing the measurement
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            probe = torch.zeros(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            backend = "nccl"
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("bench: RCCL unavailable (%s); barrier over gloo\n" % str(e)[:200])
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")
            backend = "gloo"

    import svhip
    from svhip import synth

    W, H, D, paths = CONFIGS[args.config]
    B = args.batch if args.config != "c4" else min(args.batch, 96)  # C4: 1.84 GB of volumes per pair (96 pairs = 177 GB)
    params = svhip.default_params(D, paths)  # SAD 7, P1 8*49, P2 32*49, cap 63, uniq 15, LR check on
    n_distinct = min(B, 64)   # 64 distinct pairs per stream, looped
    seed0 = 202 if world == 1 else 500 + 64 * rank
    pairs = [synth.stereo_pair(seed0 + i, W, H, D) for i in range(n_distinct)]
    lefts = [p[0] for p in pairs]
    rights = [p[1] for p in pairs]
    m = svhip.Matcher(W, H, params, device=local_rank)
    while True:   # the volumes of the default batch take 140 of the 288 GB: on a card with less to give, halve it
        try:
            m.set_batch(B)
            break
        except svhip.SvError as e:
            if B <= 64:
                raise
            sys.stderr.write("bench: batch %d does not fit (%s); trying %d\n" % (B, str(e)[:120], B // 2))
            B //= 2
    n_distinct = min(n_distinct, B)
    lt = torch.from_numpy(np.stack([lefts[i % n_distinct] for i in range(B)])).cuda()
    rt = torch.from_numpy(np.stack([rights[i % n_distinct] for i in range(B)])).cuda()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream()
    m.set_stream(stream.cuda_stream)

    step = lambda: m.compute_gray_batch_dev(B, lt, rt, out)
    if args.config == "c3":
        # full pipeline per frame: remap x2 -> gray x2 -> SGM -> vis -> WLS (16-bit map + vis) -> depth
        rng = np.random.default_rng(seed0)
        lrgb = torch.from_numpy(np.stack([synth.rgb_from_gray(rng, lefts[i % n_distinct]) for i in range(B)])).cuda()
        rrgb = torch.from_numpy(np.stack([synth.rgb_from_gray(rng, rights[i % n_distinct]) for i in range(B)])).cuda()
        cams = [synth.camera(W, H, i) for i in (0, 1)]
        for i in (0, 1):   # the synthetic pairs are already row-aligned: both views get the left camera's model
            m.init_rectify_map(i, *cams[0])
        m.update_mapper(0, *cams[0])
        fpx, base = 0.7 * W, 0.068
        Q = np.array([[1, 0, 0, -(W / 2 - 0.5)], [0, 1, 0, -(H / 2 - 0.5)], [0, 0, 0, fpx], [0, 0, 1 / base, 0]], np.float64)
        rect = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
        vis = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
        w16, w8 = torch.empty_like(out), torch.empty_like(vis)
        dep = torch.empty((B, H, W), dtype=torch.float32, device="cuda")

        def step():
            m.compute_rgb_batch_dev(B, lrgb, rrgb, out, rectify=True, rect_left_t=rect)
            m.disparity_vis_batch_dev(B, out, vis, 3)
            m.filter_wls_dev(B, out, rect, w16, vis, w8)
            m.depth_batch_dev(B, w16, Q, cams[0][2], dep)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    m.enable_timing(True)  # HIP events on the kernels' stream, read back after the timed region
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    stage_ms = m.timings()
    m.enable_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity: the timed output must be a plausible disparity map (not a skipped launch)
    valid = float((out[:, :, D:] >= 0).float().mean().item())
    if not valid > 0.5:
        raise SystemExit("bench: output looks wrong (valid fraction %.3f)" % valid)

    # what a plain copy kernel gets out of this GPU's HBM (report it beside the nominal peak)
    copy_gbs = None
    if rank == 0:
        try:
            m.synchronize()
            a_, b_ = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"), torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                b_.copy_(a_)
            e0.record()
            for _ in range(10):
                b_.copy_(a_)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = round(10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a_, b_
        except Exception:
            copy_gbs = None

    if rank == 0:
        frames = world * B * args.steps
        fps = frames / elapsed
        V = (W - D) * H * D * 2
        n_paths = bin(params.path_mask).count("1")
        n_launch, n_sweeps = m.last_aggregation()
        kernel = "svk::k_sweep" if n_sweeps == n_launch and n_sweeps > 0 else "svk::k_agg"
        launch_s = stage_ms["aggregation"] * 1e-3 / n_launch          # mean duration of one aggregation launch
        # algorithmic bytes: every path reads C and reads+writes S, all frames of the batch
        agg_bytes_per_launch = 3 * V * B * n_paths // n_launch
        achieved = agg_bytes_per_launch / launch_s / 1e9
        traffic = pmc_traffic(args.config, B, kernel)
        frame_bytes = m.algorithmic_bytes(4)
        res = {
            "metric": "disparity frames/sec per GPU at 640x480, D=128, 8 paths; % HBM roofline",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": ("%s: %dx%d D=%d %d-path SGM + LR check + median, bit-exact variant SGBM8_HH"
                                    + (" inside the full pipeline: rectify remap x2 + gray x2 + SGM + vis + WLS x2 + depth" if args.config == "c3" else ""))
                                   % (args.config, W, H, D, paths),
                       "batch_per_gpu": B, "frames_per_step": world * B, "parallelism": "frame-sharded x%d" % world,
                       "valid_fraction": round(valid, 4)},
            "per_gpu_fps": round(fps / world, 2),
            "frame_roofline_frac": round(frame_bytes * (fps / world) / (HBM_PEAK_GBS * 1e9), 4),
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_us": round(launch_s * 1e6, 2), "bytes_per_launch": agg_bytes_per_launch,
                         "paths_per_launch": n_paths // n_launch,
                         "hbm_actual_gbs": round(traffic / launch_s / 1e9, 1) if traffic else None,
                         "hbm_copy_gbs": copy_gbs},
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(W, H, D, paths, lefts, rights)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    m.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
