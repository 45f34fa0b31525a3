"""Exploration (GPU box): the 31x31 VALU convolution at 7680x4320 by chunk height (rows_per_chunk; the kernel rounds it up to whole
32-row steps): how much do the ring fill at the start of every chunk and partial rounds of the 256 one-per-CU workgroups cost?"""
import sys
sys.path.insert(0, ".")
import bench
import reforge_amd as rf
ctx = rf.Context(0)
wl = bench.WORKLOADS["conv31_8k"]
for rpc in (0, 128, 192, 256, 288, 544, 1088, 2176, 4320):
    g = rf.Graph(ctx, rf.Config(wl["text"]), wl["W"], wl["H"], wl["fmt"], conv_path=3, rows_per_chunk=rpc)
    g.fill_synthetic(wl["seed"])
    g.execute(); g.wait()
    ms = sorted(g.time_frames(40) / 40 for _ in range(3))
    r = ((rpc + 31) // 32 * 32) or 256
    chunks = (wl["H"] + r - 1) // r
    print("rows_per_chunk %4d: %d chunks x 60 strips = %4d workgroups = %.2f rounds of 256: %.4f ms" % (rpc, chunks, chunks * 60, chunks * 60 / 256.0, ms[0]), flush=True)
    g.close()
