"""Exploration (GPU box): the cache-cold rate of a workload -- frames rotating over N frame slots (own images each) so that
nothing is re-touched inside the 256 MiB Infinity Cache -- on ONE queue (rf_graph_time_frames_rotating) and with every slot on
its own stream (frames in flight, rf_graph_execute round-robin), by chunk height.
usage: cold_probe.py name:WxH[:rpc,rpc,...] ...   (names: walk_probe.TEXTS; env COLD_SLOTS, WALK_EX, WALK_T as walk_probe)"""
import os
import sys
import time

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

ctx = rf.Context(0)
TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output"}
NS = int(os.environ.get("COLD_SLOTS", "5"))
for sp in sys.argv[1:]:
    parts = sp.split(":")
    name, dims = parts[0], parts[1]
    fmt = 1
    if name.endswith("_u8"):
        name, fmt = name[:-3], 0
    rpcs = [int(x) for x in parts[2].split(",")] if len(parts) > 2 else [0]
    W, H = map(int, dims.split("x"))
    exs = {"alt": rf.RF_EXEC_ALTERNATE, "fwd": rf.RF_EXEC_NO_ALTERNATE}.get(os.environ.get("WALK_EX", ""), 0)
    t = int(os.environ.get("WALK_T", "0"))
    line = []
    for rpc in rpcs:
        g1 = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, fmt, rows_per_chunk=rpc, exec_flags=exs, texels_per_lane=t)
        g1.fill_synthetic(1)
        g1.execute(); g1.wait()
        n = max(10, int(100 / max(g1.time_frames(3) / 3, 0.02)))
        warm = min(g1.time_frames(n) / n for _ in range(3))
        g1.close()
        g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, fmt, num_frames=NS, rows_per_chunk=rpc, exec_flags=exs, texels_per_lane=t)
        g.fill_synthetic(1)
        g.time_frames_rotating(2 * NS)
        cold = min(g.time_frames_rotating(n) / n for _ in range(3))
        # frames in flight: every slot on its own stream
        for i in range(2 * NS):
            g.execute(i % NS)
        ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(n):
                g.execute(i % NS)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3 / n)
        g.close()
        line.append("rpc%d: warm %.4f cold %.4f inflight %.4f" % (rpc, warm, cold, best))
    print(name, "u8" if fmt == 0 else "f32", dims, "slots", NS, " | ".join(line), "| GB/s cold best %.0f" % (2 * W * H * (4 if fmt == 0 else 16) / (min(float(x.split()[4]) for x in line) * 1e-3) / 1e9), flush=True)
