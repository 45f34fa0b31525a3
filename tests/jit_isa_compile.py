"""Not a test: compiles the generated launches of tests/test_jit_isa.py into RF_JIT_CACHE_DIR (no device needed).
Run as a program -- `python tests/jit_isa_compile.py <cache dir> [torch]` -- it does so in a process of its own; with `torch`
PyTorch is imported FIRST, so that libhiprtc resolves to the copy PyTorch bundles (another compiler build than /opt/rocm's:
the one a pytest session or bench.py --gpus N hands the run-time compiler)."""
import os
import sys

KINDS = ["passthrough {}", "gaussian5 { sigma: 1.0 }", "gaussian9 { sigma: 2.0 }", "gaussian { sigma: 1.2, radius: %d }",
         "colour_grade { slope: 1.1, offset: 0.0, saturation: 0.9 }", "sharpen { amount: 0.5 }"]


def chain_text(rng, n):
    names = ["n%02d" % i for i in range(n)]
    decl = []
    for nm in names:
        k = KINDS[rng.randint(len(KINDS))]
        decl.append("%s: %s" % (nm, (k % rng.randint(0, 4)) if "%d" in k else k))
    return "input -> " + " -> ".join(names) + " -> output\n" + "\n".join(decl)


def fork_text(rng):
    def branch(tag):
        n = int(rng.randint(0, 3))
        names = ["%s%d" % (tag, i) for i in range(n)]
        decl = []
        for nm in names:
            k = KINDS[1 + rng.randint(len(KINDS) - 1)]
            decl.append("%s: %s" % (nm, (k % rng.randint(1, 3)) if "%d" in k else k))
        return names, decl
    a, da = branch("a")
    b, db = branch("b")
    if not a and not b:
        a, da = ["a0"], ["a0: sharpen { amount: 0.7 }"]
    pre = ["p0"] if rng.randint(2) else []
    post = ["q0"] if rng.randint(2) else []
    src = pre[-1] if pre else "input"
    lines = []
    if pre:
        lines.append("input -> p0")
    lines.append(" -> ".join([src] + a + ["mx:input_image0"]))
    lines.append(" -> ".join([src] + b + ["mx:input_image1"]))
    lines.append(" -> ".join(["mx"] + post + ["output"]))
    decl = da + db + ["mx: combination { mix: 0.3 }"] + (["p0: gaussian5 { sigma: 0.9 }"] if pre else []) + (["q0: colour_grade { slope: 1.0, offset: 0.0, saturation: 1.1 }"] if post else [])
    return "\n".join(lines + decl)


USER_TEXTS = ["input -> blur -> edges -> neg -> output\nblur: gaussian5 { sigma: 1.0 }\nedges: edge_detect { scale: 0.5 }\nneg: invert { enabled: true, strength: 1.0 }",
              "input -> ee -> output\nee: edge_detect { scale: 2.0 }"]


def compile_all(rf, shaders_dir):
    """compile ~50 generated launches (chains, fork/joins, user stages) for both formats; returns how many needed the compiler"""
    import numpy as np
    rng = np.random.RandomState(20261004)
    texts = [chain_text(rng, int(rng.randint(2, 6))) for _ in range(22)] + [fork_text(rng) for _ in range(12)]
    old = rf.shader_path()
    rf.set_shader_path(shaders_dir)
    try:
        n_jit = 0
        for k, text in enumerate(texts + USER_TEXTS):
            p = rf.Plan(rf.Config(text))
            if not any(p.needs_jit()):
                continue
            n_jit += 1
            p.jit_compile(rf.RF_FORMAT_RGBA32F)
            if k % 3 == 0:
                p.jit_compile(rf.RF_FORMAT_RGBA8)
    finally:
        rf.set_shader_path(old)
    return n_jit


if __name__ == "__main__":
    os.environ["RF_JIT_CACHE_DIR"] = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "torch":
        import torch  # noqa: F401  (first: its libhiprtc / libamd_comgr are the ones the process then uses)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import reforge_amd as rf
    n = compile_all(rf, os.path.join(root, "shaders"))
    print("compiled", n, "launches with", rf.lib().rf_jit_library().decode() if hasattr(rf.lib(), "rf_jit_library") else "?", flush=True)
    os._exit(0)
