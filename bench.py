#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:
  Mpixels/s of the 3-stage rgba32f chain (gaussian5 -> colour_grade -> sharpen) on a
  3840x2160 frame, one frame per step, plus the HBM-roofline fraction of the dominant
  kernel and the CPU oracle timed beside it.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One process per GPU.  N > 1 is WEAK scaling: rank r owns a 3840x2160 row strip of a
3840x(2160*N) frame (reforge's frames shard as row strips).  By default the
strips carry their own halo (over-fetch: the chain's 3 ghost rows are generated with the
strip, so a step needs no communication); `--halo exchange` runs the per-launch RCCL
neighbour exchange instead.

A step is one pass of the hot path over one batch of `--frames-per-step` (default 8)
synthetic frames: each frame is one rf_graph_execute = one full pass of the whole graph
over the whole frame (nothing is cached between frames); batching only keeps the timed
region long against the closing barrier when K is small.  The timed region contains
exactly K steps on inputs already resident in HBM, bracketed by a barrier + device
synchronize on both sides; the slowest rank's time is the job's time.  The oracle is used
only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHAIN3 = """
input -> blur -> grade -> sharp -> output
blur:  gaussian5    { sigma: 1.0 }
grade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp: sharpen      { amount: 0.5 }
"""
CHAIN5 = """
input -> blur -> grade -> sharp -> wide -> finish -> output
blur:   gaussian5    { sigma: 1.0 }
grade:  colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp:  sharpen      { amount: 0.5 }
wide:   gaussian9    { sigma: 2.0 }
finish: colour_grade { slope: 0.95, offset: 0.01, saturation: 0.9 }
"""

# name -> (config text, W, H per GPU, nodes, seed, description, strong?)
WORKLOADS = {
    "chain3_4k": (CHAIN3, 3840, 2160, 3, 0x5EED0002,
                  "BASELINE configs[1]: gaussian5 -> colour_grade -> sharpen, 3840x2160 rgba32f", False),
    "gauss9_8k": ("input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }", 7680, 4320, 1, 0x5EED0003,
                  "BASELINE configs[2]: 9x9 separable gaussian, 7680x4320 rgba32f", False),
    "chain5_16k": (CHAIN5, 16384, 16384, 5, 0x5EED0004,
                   "BASELINE configs[3]: 5-stage chain, 16384x16384 rgba32f, row strips (STRONG scaling)", True),
    "conv31_8k": ("input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }", 7680, 4320, 1, 0x5EED0005,
                  "BASELINE configs[4]: 31x31 dense convolution, 7680x4320 rgba32f", False),
}

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak (spec)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 MFMA peak
BPP = 16                       # rgba32f


def cpu_baseline(text, W, H, seed, budget_s=12.0):
    """The oracle (oracle/rf_oracle.c, a scalar port of one-invocation-per-pixel,
    one-pass-per-node execution) on the host cores, bounded to ~budget_s seconds per leg."""
    from oracle import graph as ograph
    from oracle import pixel

    def leg(threads, rows):
        pixel.set_threads(threads)
        g = ograph.GraphOracle(text, W, rows, pixel.FMT_RGBA32F)
        g.upload_raw(pixel.fill_synthetic(W, rows, pixel.FMT_RGBA32F, seed))
        t0 = time.perf_counter()
        g.execute()
        first = time.perf_counter() - t0
        reps = max(1, min(50, int(budget_s / max(first, 1e-3)) - 1))
        t0 = time.perf_counter()
        for _ in range(reps):
            g.execute()
        dt = (time.perf_counter() - t0) / reps
        return W * rows / dt / 1e6, reps

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))      # the GPU box's CPU share for one GPU
    rows = min(H, 2160)
    all_v, all_reps = leg(cores, rows)
    one_v, one_reps = leg(1, min(rows, 540))
    pixel.set_threads(1)
    import ctypes.util
    vk = ctypes.util.find_library("vulkan")     # the reference itself would need a Vulkan loader + lavapipe
    return {
        "value": round(all_v, 2), "unit": "Mpx/s", "cores": cores, "kind": "port",
        "sample": "%d x %d rgba32f frame, whole chain, %d repetitions after one warm-up (oracle/rf_oracle.c, OpenMP over rows)" % (W, rows, all_reps),
        "single_thread_value": round(one_v, 2),
        "single_thread_sample": "%d x %d rows, %d repetitions" % (W, min(rows, 540), one_reps),
        "lavapipe": "unavailable (no libvulkan on this box)" if vk is None else "libvulkan present (%s), no reference build to drive it" % vk,
    }


def load_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json),
    null when no profile of this workload exists."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as fh:
            return json.load(fh).get(key)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="chain3_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--halo", default="overfetch", choices=["overfetch", "exchange"])
    ap.add_argument("--no-fusion", action="store_true", help="one launch per node, as the reference dispatches")
    ap.add_argument("--hipgraph", action="store_true")
    ap.add_argument("--frames-per-step", type=int, default=8,
                    help="frames in the batch one step processes (each frame = one full pass of the graph)")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="frame slots the batch alternates over (reforge's --num-frames; each slot has its own stream and images)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 plumbing check on a one-GPU box: all ranks on device 0, gloo barrier; numbers are meaningless")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))

    import torch

    import reforge_amd as rf

    if args.rehearse:
        # plumbing rehearsal on a one-GPU box: every rank on device 0, gloo for the barrier
        # (RCCL refuses two ranks on one device).  The numbers it prints mean nothing.
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if args.rehearse else "cuda"

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    text, W, Hper, n_nodes, seed, desc, strong = WORKLOADS[args.workload]
    H = Hper if (strong or world == 1) else Hper * world

    uid = None
    if world > 1 and args.halo == "exchange":
        t = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            t = torch.frombuffer(bytearray(rf.Context.unique_id()), dtype=torch.uint8).to(red_dev)
        dist.broadcast(t, 0)
        uid = bytes(t.cpu().numpy().tobytes())
    ctx = rf.Context(local_rank, rank, world, uid) if world > 1 else rf.Context(local_rank)

    flags = 0
    if args.no_fusion:
        flags |= rf.RF_GRAPH_NO_FUSION
    if args.hipgraph:
        flags |= rf.RF_GRAPH_HIPGRAPH
    if world > 1 and args.halo == "overfetch":
        flags |= rf.RF_GRAPH_NO_HALO_XCHG
    nslots = max(1, args.frames_in_flight)
    g = rf.Graph(ctx, rf.Config(text), W, H, rf.RF_FORMAT_RGBA32F, num_frames=nslots, flags=flags)
    g.fill_synthetic(seed)                      # inputs resident in HBM before anything is timed
    launches = g.plan.launch_info()

    for i in range(args.warmup * args.frames_per_step):
        g.execute(i % nslots)
    for sl in range(nslots):
        g.wait(sl)

    # ---- the timed region: exactly K steps ------------------------------------------------
    barrier_sync()
    t0 = time.perf_counter()
    fps = args.frames_per_step
    for _ in range(args.steps):
        for i in range(fps):                           # one step = one batch of `fps` frames
            g.execute(i % nslots)
    ctx.synchronize()
    barrier_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = elapsed / args.steps * 1e3
    total_px = W * H                                   # the whole job's frame
    value = total_px * fps / (elapsed / args.steps) / 1e6

    # ---- roofline of the dominant kernel: HIP events on the launch's own stream -------------
    rows = g.rows
    # One launch per frame: two events around a run of back-to-back frames on the launch's
    # stream (an event pair around EVERY launch would put a marker packet between kernels and
    # inflate each by ~2 us).  Several launches per frame: a pair per launch, to tell them apart.
    n_ev = max(20, min(args.steps * fps, 400))
    if len(launches) == 1:
        per_launch = [(launches[0]["label"], g.time_frames(n_ev) / n_ev)]
    else:
        per_launch = g.time_launches(min(n_ev, 100))
    dom = max(range(len(per_launch)), key=lambda i: per_launch[i][1])
    dom_label, dom_ms = per_launch[dom]
    if args.workload == "conv31_8k":
        flops = 2.0 * 961 * 4 * W * rows
        achieved = flops / (dom_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": load_traffic(args.workload)}
    else:
        # algorithmic bytes of ONE launch: (inputs + outputs) x W x rows x 16 B.  A fused launch
        # is priced as the single read + single write it performs (32 B/px), NOT as the sum
        # of the nodes it covers (that figure is reported as chain_hbm_frac below).
        n_in = len(launches[dom]["inputs"])
        alg_bytes = (n_in + 1) * W * rows * BPP
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": load_traffic(args.workload + ("_unfused" if args.no_fusion else ""))}
    each = sorted(g.time_each_frame(max(20, min(args.steps * fps, 200))))     # median and min per frame
    frame_events = {"median_ms": round(each[len(each) // 2], 5), "min_ms": round(each[0], 5), "frames": len(each),
                    "note": "hipEvent pair per frame on the frame's stream (adds a marker packet per frame)"}
    roofline["kernel"] = dom_label
    roofline["launch_ms"] = round(dom_ms, 5)
    roofline["algorithmic_bytes_per_px"] = 32 if roofline["bound"] == "hbm" else None

    # BASELINE.md's "% HBM roofline": the per-node algorithmic bytes of the whole chain
    chain_bytes = n_nodes * 2 * BPP * total_px
    chain_frac = chain_bytes / (elapsed / args.steps / fps) / 1e9 / (HBM_PEAK_GBS * world)

    out = {
        "metric": "Mpixels/sec, 3-stage rgba32f chain @4K" if args.workload == "chain3_4k" else "Mpixels/sec, " + args.workload,
        "value": round(value, 1),
        "unit": "Mpx/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,                           # BASELINE.md holds no published number
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "frame": "%dx%d" % (W, H), "rows_per_gpu": rows, "format": "rgba32f",
            "frames_per_step": fps, "ms_per_frame": round(ms_per_step / fps, 5),
            "nodes": n_nodes, "launches_per_frame": len(launches), "launches": [l["label"] for l in launches],
            "fusion": not args.no_fusion, "hipgraph": bool(args.hipgraph), "frames_in_flight": nslots,
            "parallelism": "1 GPU" if world == 1 else "row strips x%d, halo=%s" % (world, args.halo),
        },
        "roofline": roofline,
        "chain_hbm_frac": round(chain_frac, 4),
        "chain_algorithmic_bytes_per_px": n_nodes * 2 * BPP,
        "launch_ms": {k: round(v, 5) for k, v in per_launch},
        "frame_ms_events": frame_events,
    }
    if args.workload == "conv31_8k" and world == 1:
        # both large-K kernels, timed the same way (the default is the VALU one; RF_CONV_PATH=2 selects MFMA)
        paths = {}
        for name, env in (("valu", "3"), ("mfma", "2")):
            os.environ["RF_CONV_PATH"] = env
            g2 = rf.Graph(ctx, rf.Config(text), W, H, rf.RF_FORMAT_RGBA32F, num_frames=1, flags=flags)
            g2.fill_synthetic(seed)
            g2.execute(0)
            g2.wait(0)
            ms = g2.time_frames(5) / 5
            paths[name] = {"ms": round(ms, 4), "useful_tflops": round(2.0 * 961 * 4 * W * rows / (ms * 1e-3) / 1e12, 2)}
            g2.close()
        del os.environ["RF_CONV_PATH"]
        out["conv_kernels"] = paths
    if rank == 0 and world == 1:
        try:
            out["copy_gbps"] = round(ctx.copy_bandwidth(256 << 20, 20), 1)
        except rf.RfError:
            out["copy_gbps"] = None
        if not args.skip_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(text, W, H, seed)
    g.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
