"""Exploration (GPU box, under rocprofv3 --pmc): a few 8K 31x31 conv2d frames on the path RF_CONV_PATH names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import reforge_amd as rf

ctx = rf.Context(0)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
g = rf.Graph(ctx, rf.Config("input -> conv2d -> output\nconv2d: conv2d { ksize: 31, sigma: 5.0 }"), W, H, rf.RF_FORMAT_RGBA32F)
g.fill_synthetic(5)
for _ in range(3):
    g.execute()
g.wait()
g.close()
