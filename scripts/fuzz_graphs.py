"""Exploration (GPU box): many generated graphs at a mid-size frame against the oracle.
usage: fuzz_graphs.py <first seed> <count> [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import reforge_amd as rf
from oracle import pixel
from tests import util

first, count = int(sys.argv[1]), int(sys.argv[2])
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
ctx = rf.Context(0)
if os.environ.get("FUZZ_GLSL") == "1":
    # the node types that have a bit-exact GLSL twin run FROM THEIR .comp FILES (file-first lookup, rf_glsl.cpp), the oracle runs its own
    # restatement of the same types: needs FUZZ_GEN=dag_user.  (Not the gaussians -- a shader derives their weights with GLSL exp() -- nor
    # conv2d / colour_grade, whose files lack the built-ins' optional bindings.)
    import shutil, tempfile
    gd = tempfile.mkdtemp(prefix="fuzz_glsl_")
    for t in ("sharpen", "combination", "split_luma", "invert", "edge_detect", "unsharp_mask", "local_contrast"):
        shutil.copy(os.path.join(util.SHADERS, t + ".comp"), gd)
    for t in ("tone_curve", "apply_curve", "streak"):      # user types without a twin: their stage files
        shutil.copy(os.path.join(util.SHADERS, t + ".stage.hip"), gd)
pixel.set_threads(min(16, os.cpu_count() or 1))
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    gen = os.environ.get("FUZZ_GEN")
    if gen == "dag_user":      # user types (shaders/*.stage.hip) mixed in; the oracle compiles the same files for the host
        if seed == first:
            util.register_user_types()
            if os.environ.get("FUZZ_GLSL") == "1":
                rf.set_shader_path(gd)
                rf.set_type_lookup(True)
        text = util.random_dag(rng, split=seed % 2 == 0, user=True)
    else:
        text = (util.random_dag if gen == "dag" else util.random_graph)(rng)
    w = W if seed % 3 else int(rng.randint(1, W))
    h = H if seed % 5 else int(rng.randint(1, H))
    for fmt in (util.F32, util.U8):
        x = pixel.fill_synthetic(w, h, fmt, seed)
        try:
            want = util.run_oracle(text, x)
        except Exception as e:
            print("seed", seed, "oracle failed:", e, "\n" + text, flush=True)
            bad += 1
            break
        for flags in (0, rf.RF_GRAPH_NO_FUSION, rf.RF_GRAPH_HIPGRAPH, rf.RF_GRAPH_NO_JIT):
            try:
                got = util.run_hip(ctx, text, x, flags=flags)
                util.assert_same(got, want, "")
            except Exception as e:
                bad += 1
                print("seed", seed, "fmt", fmt, "flags", flags, "%dx%d" % (w, h), str(e)[:300], "\n" + text, flush=True)
    if (seed - first) % 20 == 19:
        print("progress", seed - first + 1, "graphs,", bad, "failures, %.0f s, %d kernels compiled" % (time.time() - t0, rf.lib().rf_jit_compile_count()), flush=True)
print("done", count, "graphs,", bad, "failures, %.0f s" % (time.time() - t0), flush=True)
