"""Exploration (GPU box): the dynamic tail of the stream launches against the static schedule, INTERLEAVED on one box.
usage: steal_probe.py name:WxH[:variant,variant,...] ...      variant = static | dyn | dynR<rounds>[uN][cROWS] | statC<rows>
Every variant is a graph of its own (the knobs are read at rf_graph_create); `rounds` passes over all variants, the best and
the median of the per-pass averages are printed, with the walks taken over per frame."""
import os
import statistics
import sys

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

ctx = rf.Context(0)
TEXTS = {"chain5": bench.CHAIN5, "chain3": bench.CHAIN3, "gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "pass": "input -> passthrough -> output",
         "sharpen": "input -> sh -> output\nsh: sharpen { amount: 0.5 }", "gauss5": "input -> blur -> output\nblur: gaussian5 { sigma: 1.0 }",
         "diamond": bench.WORKLOADS["diamond_4k"]["text"]}
ROUNDS = int(os.environ.get("PROBE_ROUNDS", "3"))


def make(name, W, H, fmt, variant):
    env = {}
    kw = {}
    if variant == "static":
        kw["exec_flags"] = rf.RF_EXEC_STATIC_WALKS
    elif variant.startswith("statC"):
        kw["exec_flags"] = rf.RF_EXEC_STATIC_WALKS
        kw["rows_per_chunk"] = int(variant[5:])
    elif variant.startswith("dyn"):
        rest = variant[3:]
        if rest.startswith("R"):
            num = ""
            rest = rest[1:]
            while rest and rest[0].isdigit():
                num, rest = num + rest[0], rest[1:]
            env["RF_STEAL_ROUNDS"] = num
        if rest.startswith("u"):
            num = ""
            rest = rest[1:]
            while rest and rest[0].isdigit():
                num, rest = num + rest[0], rest[1:]
            kw["walk_unit"] = int(num)
        if rest.startswith("c"):
            kw["rows_per_chunk"] = int(rest[1:])
            kw["exec_flags"] = rf.RF_EXEC_DYNAMIC_WALKS
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g = rf.Graph(ctx, rf.Config(TEXTS[name]), W, H, fmt, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    g.fill_synthetic(1)
    g.execute()
    g.wait()
    return g


for sp in sys.argv[1:]:
    parts = sp.split(":")
    name, dims = parts[0], parts[1]
    fmt = 1
    if name.endswith("_u8"):
        name, fmt = name[:-3], 0
    variants = parts[2].split(",") if len(parts) > 2 else ["static", "dyn"]
    W, H = map(int, dims.split("x"))
    # ONE graph alive at a time: hipMalloc then hands every variant the same image addresses -- where the images lie moves a
    # beyond-cache launch by up to 9 % (scripts/placement_probe.py, profiles/r04_placement.txt), more than most variants differ
    res = {v: [] for v in variants}
    taken = {v: 0 for v in variants}
    frames = {v: 0 for v in variants}
    n = 0
    for _ in range(ROUNDS):
        for v in variants:
            g = make(name, W, H, fmt, v)
            if n == 0:
                n = max(4, int(30 / max(g.time_frames(2) / 2, 0.02)))
            t0 = g.walks_taken()
            res[v].append(min(g.time_frames(n) / n for _ in range(2)))
            taken[v] += g.walks_taken() - t0
            frames[v] += 2 * n
            g.close()
    base = min(res[variants[0]])
    print("%s %s %s  (%d frames per pass, %d passes)" % (name, "u8" if fmt == 0 else "f32", dims, n, ROUNDS), flush=True)
    for v in variants:
        print("   %-14s best %.4f ms  median %.4f  (%+.1f %% vs %s)  walks taken / frame %.0f" %
              (v, min(res[v]), statistics.median(res[v]), 100.0 * (min(res[v]) / base - 1.0), variants[0], taken[v] / max(frames[v], 1)), flush=True)
