"""Not a test: compiles a few shaders/*.comp graphs into RF_JIT_CACHE_DIR (no device needed), for tests/test_glsl.py.
`python tests/glsl_isa_compile.py <cache dir> [torch]`: with `torch` PyTorch is imported FIRST, so that libhiprtc resolves to the copy
PyTorch bundles (another compiler build: the one bench.py and a pytest session that imported torch hand the run-time compiler)."""
import os
import shutil
import sys
import tempfile

GRAPHS = {
    "gaussian5": "input -> gg -> output\ngg: gaussian5 { sigma: 1.0 }",
    "local_contrast": "input -> lc -> output\nlc: local_contrast { amount: 0.8 }",
    "sharpen": "input -> sh -> output\nsh: sharpen { amount: 0.5 }",
    "colour_grade": "input -> cg -> output\ncg: colour_grade { slope: 1.1 }",
    "unsharp_mask": "input -> bl -> um:blurred_image\ninput -> um:input_image\num -> output\nbl: passthrough {}\num: unsharp_mask { amount: 1.5 }",
}


def compile_all(rf, root):
    d = tempfile.mkdtemp(prefix="rf_glsl_isa_")
    for t in GRAPHS:
        shutil.copy(os.path.join(root, "shaders", t + ".comp"), d)
    old = rf.shader_path()
    rf.set_shader_path(d)
    rf.set_type_lookup(True)
    try:
        for text in GRAPHS.values():
            for flags in (0, rf.RF_GRAPH_GLSL_NODES):      # as planned (row stages where a file is one), and every file as a node with a kernel of its own
                p = rf.Plan(rf.Config(text), flags)
                p.jit_compile(rf.RF_FORMAT_RGBA32F)
                p.jit_compile(rf.RF_FORMAT_RGBA8)
    finally:
        rf.set_type_lookup(False)
        rf.set_shader_path(old)
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    os.environ["RF_JIT_CACHE_DIR"] = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "torch":
        import torch  # noqa: F401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import reforge_amd as rf
    compile_all(rf, root)
    print("compiled with", rf.lib().rf_jit_library().decode(), flush=True)
    os._exit(0)
