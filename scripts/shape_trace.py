"""Exploration (GPU box): print the launch geometry librfhip chooses (RF_TRACE_SHAPE=1) for a list of workloads."""
import sys
sys.path.insert(0, ".")
import bench, reforge_amd as rf
ctx = rf.Context(0)
T = {"gauss9": bench.WORKLOADS["gauss9_8k"]["text"], "chain3": bench.CHAIN3, "chain5": bench.CHAIN5, "sharpen": "input -> sh -> output\nsh: sharpen { amount: 0.5 }", "pass": "input -> passthrough -> output"}
CASES = (("gauss9", 7680, 4320, 0), ("chain3", 7680, 4320, 0), ("sharpen", 7680, 4320, 0), ("sharpen", 7680, 4320, 1), ("gauss9", 7680, 4320, 1), ("chain3", 3840, 2160, 1), ("chain3", 3840, 2160, 0), ("chain5", 16384, 16384, 1), ("pass", 3840, 2160, 1), ("chain5", 7680, 4320, 1), ("chain3", 7680, 4320, 1), ("chain5", 3840, 2160, 1), ("chain5", 1920, 1080, 1), ("chain3", 1920, 1080, 1),
         ("chain5", 16384, 2048, 1), ("chain5", 16384, 8192, 1), ("chain5", 16384, 4096, 1))
for name, W, H, fmt in CASES:
    print(name, W, H, "u8" if fmt == 0 else "f32", flush=True)
    g = rf.Graph(ctx, rf.Config(T[name]), W, H, fmt)
    g.fill_synthetic(1); g.execute(); g.wait(); g.close()
