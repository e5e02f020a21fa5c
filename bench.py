#!/usr/bin/env python3
"""bench.py — disparity frames/s per GPU at 640x480, D=128, 8 paths (+ LR check), % of HBM roofline.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1: plain process.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL is used
  only for the timing barrier / max-over-ranks; stereo pairs are independent, so frames are
  sharded across ranks with NO data-path collective -> "scaling": "weak").

A "step" is one pass of the hot path (prefilter -> cost volume -> 8-path SGM -> WTA -> LR check ->
median; sv_compute_gray_batch_dev) over one batch of synthetic stereo pairs that are already
resident in HBM.  value = frames of all ranks / max-over-ranks wall time of exactly K steps.

roofline: the dominant kernel is the aggregation launch that takes longest: with the fused sweeps (large
batches, D <= 256) svk::k_sweep<N, true, CW>, the second sweep (paths L, UL, U, UR + winner-take-all); with
per-path launches svk::k_agg.  Its launch duration is measured live with HIP events that libsv_hip records
on the stream the kernels run on, after EVERY aggregation launch, over the K timed steps
(sv_get_aggregation_timings).  achieved = ALGORITHMIC bytes of that launch / that duration, where the
algorithmic bytes of a fused launch are what one pass must move: sweep 1 = 2 V per frame
(C read, S written), sweep 2 = 2 V + 8 W H (C and S read, 8-byte WTA records written), cost = V + 2 W H,
V = (W-D)*H*D*2; a per-path launch = 3 V.  traffic = HBM bytes of the same kernel from the
committed rocprofv3 --pmc passes (profiles/r03_traffic.json; FETCH_SIZE x2 + WRITE_SIZE, the gfx950
correction of MI355X_MICROARCH.md), null when that file was taken from other kernel sources than the ones
this run loads.  SURVEY 8(d)'s per-path model (3 V per path whatever the fusion) is reported beside it as
per_path_model_gbs: a fused sweep moves a third of those bytes, so that figure exceeds the HBM peak and is
no roofline fraction.

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
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
           "host_cpus": os.cpu_count(),
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


KERNEL_SOURCES = ("sv_hip.hip", "sv_kernels.hip.h", "sv_sweep.hip.h", "sv_blf.hip.h")


def source_sha():
    """sha256 over the kernel sources libsv_hip.so is built from: ties a committed PMC pass to a build."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "stereo-vision_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(config, batch):
    """{kernel name: HBM bytes per launch} from the committed rocprofv3 PMC passes (profiles/r03_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this workload, gfx950 FETCH_SIZE x2 correction applied), the
    file's name and the sources it belongs to; ({}, ...) when no pass exists for this config / batch or the pass
    was taken from other kernel sources than the ones in the tree now."""
    path = os.path.join("profiles", "r03_traffic.json")
    try:
        with open(os.path.join(ROOT, path)) as f:
            t = json.load(f)
        entry = t.get("%s_b%d" % (config, batch))
        if not entry or t.get("source_sha") != source_sha():
            return {}, path, t.get("source_sha")
        return {k: int(v["traffic_bytes_per_launch"]) for k, v in entry["kernels"].items()}, path, t.get("source_sha")
    except (OSError, ValueError, KeyError, TypeError):
        return {}, path, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="(default 1536; c4: 112) stereo pairs per step per GPU (C2: 1536 pairs = 213 GB of the 288 GB of HBM; every workgroup pipeline then streams 48 frames and the fill/drain of the band pipeline is 3.4 %% of a launch: 5 %% at 1024, 10 %% at 512)")
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
    dev = local_rank
    if os.environ.get("SV_BENCH_SHARE_DEVICE") == "1":
        # rehearsal of the world > 1 branch on a box with fewer GPUs than ranks (the ranks then share devices; RCCL refuses
        # that and the barrier goes over gloo): exercises rank seeds, the barrier and the max-over-ranks rule, not scaling
        dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    backend = None
    if world > 1:
        # RCCL carries only the timing barrier and the max-over-ranks; if it cannot come up on this node the
        # same two operations go over gloo (host tensors) rather than losing the measurement
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
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
    # C4: 1.84 GB of volumes per pair: 112 pairs = 206 GB = 7 pipelines x 16 frames (a pipeline's block of volumes has to stay
    # below 16 GiB for the sweeps' 32-bit offsets: 18 frames per pipeline would split the launch)
    B = args.batch if args.batch else (112 if args.config == "c4" else 1536)
    params = svhip.default_params(D, paths)  # SAD 7, P1 8*49, P2 32*49, cap 63, uniq 15, LR check on
    n_distinct = min(B, 64)   # 64 distinct pairs per stream, looped
    seed0 = 202 if world == 1 else 500 + 64 * rank
    pairs = [synth.stereo_pair(seed0 + i, W, H, D) for i in range(n_distinct)]
    lefts = [p[0] for p in pairs]
    rights = [p[1] for p in pairs]
    m = svhip.Matcher(W, H, params, device=dev)
    while True:   # the volumes of the default batch take 213 of the 288 GB: on a card with less to give, step down
        try:
            m.set_batch(B)
            break
        except svhip.SvError as e:
            if B <= 64:
                raise
            nxt = 1024 if B > 1024 else B // 2
            sys.stderr.write("bench: batch %d does not fit (%s); trying %d\n" % (B, str(e)[:120], nxt))
            B = nxt
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
        lsrc = [synth.rgb_from_gray(rng, lefts[i]) for i in range(n_distinct)]
        rsrc = [synth.rgb_from_gray(rng, rights[i]) for i in range(n_distinct)]
        cams = [synth.camera(W, H, i) for i in (0, 1)]
        # the synthetic pairs are row-aligned: what gets rectified is the two cameras' raw views of them
        raw = [synth.raw_pair(lsrc[i], rsrc[i], cams) for i in range(n_distinct)]
        lsrc, rsrc = [r_[0] for r_ in raw], [r_[1] for r_ in raw]
        lrgb = torch.from_numpy(np.stack([lsrc[i % n_distinct] for i in range(B)])).cuda()
        rrgb = torch.from_numpy(np.stack([rsrc[i % n_distinct] for i in range(B)])).cuda()
        for i in (0, 1):
            m.init_rectify_map(i, *cams[i])
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
    m.synchronize()        # raises if a warm-up launch reported a device-side error
    m.enable_timing(True)  # HIP events on the kernels' stream, read back after the timed region
    out.fill_(-32768)      # the timed steps must rewrite every pixel: nothing stale from the warm-up survives
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    # every rank, outside any try: a sweep that timed out (deferred device-side error) fails the run
    m.synchronize()
    stage_ms = m.timings()
    agg_ms = m.aggregation_timings()
    m.enable_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # The timed output must be the right answer, not just a plausible one (no oracle here: the oracle is the
    # checker of tests/ and smoke()).  (1) every pixel was rewritten and most are valid; (2) the batch loops
    # n_distinct pairs, so frame i must equal frame i mod n_distinct - frames that sit in different workgroup
    # pipelines of the sweep; (3) the first frames must equal what the per-path kernels (another code path,
    # pinned to the oracle by tests/) give for them.
    n_launch, n_sweeps = m.last_aggregation()
    if int((out == -32768).sum().item()) != 0:
        raise SystemExit("bench: the timed steps left pixels unwritten")
    valid = float((out[:, :, D:] >= 0).float().mean().item())
    if not valid > 0.5:
        raise SystemExit("bench: output looks wrong (valid fraction %.3f)" % valid)
    if B > n_distinct:
        reps = B // n_distinct
        same = bool((out[:reps * n_distinct].view(reps, n_distinct, H, W) == out[:n_distinct].unsqueeze(0)).all().item())
        if not same:
            raise SystemExit("bench: frames of the same pair differ across the batch")
    if args.config != "c3" and n_sweeps > 0:
        ncheck = min(4, B)
        chk = torch.empty((ncheck, H, W), dtype=torch.int16, device="cuda")
        m.set_debug(4)   # plain per-path launches
        m.compute_gray_batch_dev(ncheck, lt, rt, chk)
        m.synchronize()
        m.set_debug(0)
        if not bool((chk == out[:ncheck]).all().item()):
            raise SystemExit("bench: fused sweeps and per-path kernels disagree")

    # what a plain copy kernel gets out of this GPU's HBM (report it beside the nominal peak)
    copy_gbs = None
    if rank == 0:
        try:
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
        fused = n_sweeps > 0   # (a few frames beyond whole pipeline rounds may have gone through the all-paths kernels: more launches than sweeps)
        cw = m.last_sweep_waves()
        # per-launch algorithmic bytes and kernel names, in launch order
        if fused:
            names = ["svk::k_sweep<%d, false, %d>" % (D // 32, cw), "svk::k_sweep<%d, true, %d>" % (D // 32, cw)]
            alg = [2 * V * B, (2 * V + 8 * W * H) * B]
        else:
            names = ["svk::k_agg"] * len(agg_ms)
            alg = [3 * V * B] * len(agg_ms)
        traffic_by_kernel, traffic_file, traffic_sha = pmc_traffic(args.config, B)
        kstats = []
        for nm, ms_, by in zip(names, agg_ms, alg):
            tr = traffic_by_kernel.get(nm)
            kstats.append({"kernel": nm, "launch_us": round(ms_ * 1e3, 2), "algorithmic_bytes": by,
                           "achieved_gbs": round(by / (ms_ * 1e-3) / 1e9, 1), "frac": round(by / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "traffic": tr, "hbm_actual_gbs": round(tr / (ms_ * 1e-3) / 1e9, 1) if tr else None})
        # the kernel launch_cost2 picks for a gray batch with 7x7 blocks: D > 128 runs as disparity parts of the D <= 128 kernel
        n32 = D // 32
        cost_name = "svk::k_cost2<%d, %d, 1>" % ((n32 if n32 <= 4 else (n32 + 1) // 2, 1024) if B >= 8 else (n32, 512))
        cost_bytes = (V + 2 * W * H) * B
        cost_tr = traffic_by_kernel.get(cost_name)
        kstats.append({"kernel": cost_name + " (+ k_cost_col0; stage time)", "launch_us": round(stage_ms["cost"] * 1e3, 2),
                       "algorithmic_bytes": cost_bytes, "achieved_gbs": round(cost_bytes / (stage_ms["cost"] * 1e-3) / 1e9, 1),
                       "frac": round(cost_bytes / (stage_ms["cost"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": cost_tr,
                       "hbm_actual_gbs": round(cost_tr / (stage_ms["cost"] * 1e-3) / 1e9, 1) if cost_tr else None})
        dom = max(kstats[:len(agg_ms)], key=lambda k: k["launch_us"])   # the dominant kernel: the longest aggregation launch
        # whole frame: the bytes a fully fused pipeline must move (cost writes C; each sweep reads C; S written once and
        # read once; records, planes) against SURVEY 8(d)'s per-path model
        # (+ 60 bytes per pixel of images, BT statistics written and read, WTA records written and read, output)
        fused_frame_bytes = 5 * V + 60 * W * H
        frame_bytes = m.algorithmic_bytes(4)
        per_path_gbs = 3 * V * B * n_paths / (sum(agg_ms) * 1e-3) / 1e9
        res = {
            "metric": "disparity frames/sec per GPU at %dx%d, D=%d, %d paths; %% HBM roofline" % (W, H, D, paths),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": ("%s: %dx%d D=%d %d-path SGM + LR check + median, bit-exact variant SGBM8_HH"
                                    + (" inside the full pipeline: rectify remap x2 + gray x2 + SGM + vis + WLS x2 + depth" if args.config == "c3" else ""))
                                   % (args.config, W, H, D, paths),
                       "batch_per_gpu": B, "frames_per_step": world * B, "parallelism": "frame-sharded x%d" % world,
                       "valid_fraction": round(valid, 4),
                       "checked": "all pixels rewritten; frame i == frame i mod %d over the batch; first frames == per-path kernels" % n_distinct},
            "per_gpu_fps": round(fps / world, 2),
            "frame_fused_bytes_frac": round(fused_frame_bytes * (fps / world) / (HBM_PEAK_GBS * 1e9), 4),
            "frame_per_path_model_x_peak": round(frame_bytes * (fps / world) / (HBM_PEAK_GBS * 1e9), 4),
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_gbs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
                         "launch_us": dom["launch_us"], "bytes_per_launch": dom["algorithmic_bytes"],
                         "hbm_actual_gbs": dom["hbm_actual_gbs"],
                         "hbm_actual_frac": round(dom["hbm_actual_gbs"] / HBM_PEAK_GBS, 4) if dom["hbm_actual_gbs"] else None,
                         "traffic_source": {"file": traffic_file, "kernel_sources_sha": traffic_sha, "this_run_sha": source_sha()},
                         "limiter": "VALU issue and HBM together (DESIGN.md section 4); the HBM fraction is what the north star asks for",
                         # what a load-only kernel with this kernel's streams reaches on an MI355X (recorded, not measured in this run)
                         "pattern_ceiling_gbs": {"read_C_read_S": 6020, "read_C_write_S": 5680, "source": "profiles/r03_f_access_pattern_ubench.txt"},
                         "per_path_model_gbs": round(per_path_gbs, 1),
                         "hbm_copy_gbs": copy_gbs, "kernels": kstats},
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
