"""Exploration (GPU box, under rocprofv3 --pmc): a few frames of one of the BASELINE graphs.
usage: stream_pmc.py <chain3|chain5|gauss9|passthrough> <f32|u8> <W> <H> [flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf
from tests import util

TEXT = {"chain3": util.CHAIN3, "chain5": util.CHAIN5, "passthrough": "input -> passthrough -> output",
        "gauss9": "input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }",
        "grade": "input -> grade -> output\ngrade: grade { slope: 1.3, offset: -0.1, saturation: 0.6 }"}
name, fmt, W, H = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = rf.Context(0)
g = rf.Graph(ctx, rf.Config(TEXT[name]), W, H, rf.RF_FORMAT_RGBA32F if fmt == "f32" else rf.RF_FORMAT_RGBA8, flags=flags)
g.fill_synthetic(5)
for _ in range(5):
    g.execute()
g.wait()
g.close()
