"""Exploration (GPU box): one vs two texels per lane (rf_graph_options.texels_per_lane) on the BASELINE
stream workloads, interleaved rounds in one process, bit-equality of the two outputs checked on a band."""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench
import reforge_amd as rf

ctx = rf.Context(0)
cases = [("chain3_4k", 3840, 2160), ("gauss9_8k", 7680, 4320), ("chain5_16k", 16384, 16384), ("chain3_8k", 7680, 4320), ("chain5_8k", 7680, 4320)]
only = sys.argv[1:] or [c[0] for c in cases]
for name, W, H in cases:
    if name not in only:
        continue
    wl = bench.WORKLOADS.get(name) or dict(bench.WORKLOADS["chain3_4k" if "chain3" in name else "chain5_16k"])
    gs = {}
    for t in (1, 2):
        for rpc in (0,):
            g = rf.Graph(ctx, rf.Config(wl["text"]), W, H, 1, texels_per_lane=t, rows_per_chunk=rpc)
            g.fill_synthetic(wl["seed"])
            g.execute(); g.wait()
            gs[(t, rpc)] = g
    ref = None
    for key, g in gs.items():
        band = g.download_rows(H // 2 - 8, H // 2 + 8).tobytes() + g.download_rows(0, 8).tobytes() + g.download_rows(H - 8, H).tobytes()
        ref = ref or band
        assert band == ref, (name, key, "differs")
    n = max(5, int(40 / max(gs[(1, 0)].time_frames(2) / 2, 0.02)))
    res = {k: [] for k in gs}
    for rnd in range(5):
        for k, g in gs.items():
            res[k].append(g.time_frames(n) / n)
    print(name, "%dx%d" % (W, H), " ".join("T=%d rpc=%d: med %.4f min %.4f ms |" % (k[0], k[1], sorted(v)[2], min(v)) for k, v in res.items()), flush=True)
    for g in gs.values():
        g.close()
