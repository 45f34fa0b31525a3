"""Exploration (GPU box): generated graphs in EXCHANGE mode, 2..4 processes sharing GPU 0 over the RCCL
test double (tests/native/fake_rccl.cpp), against the oracle.  usage: fuzz_exchange.py <first seed> <count>"""
import os, pathlib, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util
from tests.test_gpu_exchange import run_ranks, ROOT, HIPCC

first, count = int(sys.argv[1]), int(sys.argv[2])
fake = tempfile.mkdtemp()
subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-shared", "-fPIC", "-x", "c++", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"),
                       "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "reforge_amd", "csrc"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", os.path.join(fake, "librccl.so.1")],
                      stderr=subprocess.DEVNULL)
pixel.set_threads(8)
bad = skipped = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    world = int(rng.randint(2, 5))
    W, H = int(rng.randint(20, 700)), int(rng.randint(120, 500))
    flags = (0, rf.RF_GRAPH_NO_FUSION)[(seed >> 1) & 1]
    fmt = (util.F32, util.U8)[(seed >> 2) & 1]
    source = ("fill", "upload")[(seed >> 3) & 1]
    if (seed >> 4) & 1:
        flags |= rf.RF_GRAPH_NO_HALO_XCHG
    if rf.Plan(rf.Config(text), flags & rf.RF_GRAPH_NO_FUSION).halo_schedule(not (flags & rf.RF_GRAPH_NO_HALO_XCHG))[3] > H // world:
        skipped += 1
        continue
    d = pathlib.Path(tempfile.mkdtemp())
    try:
        got = run_ranks(fake, d, text, world, W, H, fmt, flags, seed, frames=1 + (seed % 2), source=source)
        if text.startswith("input -> n00:image") or "input -> n00:image" in text:
            continue      # an in-place head on the input compounds over frames: covered elsewhere
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed))
        util.assert_same(got, want, "")
    except BaseException as e:
        bad += 1
        print("seed", seed, "world", world, "flags", flags, "fmt", fmt, source, "%dx%d" % (W, H), str(e)[:300], "\n" + text, flush=True)
print("done", count, "graphs,", skipped, "skipped,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
