"""CPU: the host-only half of the library (config parser, planner, ABI accessors) built with
AddressSanitizer + UBSan + LeakSanitizer and fed 4500 generated configs (valid graphs from both
generators and token soup) and 700 generated stage files and 900 generated GLSL files (the shipped ones mutated, token soup).  GPU sanitizers are not available on the pool; this covers the code
that handles untrusted text."""
import os
import subprocess

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reforge_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++")
def test_parser_and_planner_under_sanitizers(tmp_path):
    exe = str(tmp_path / "plan_asan")
    subprocess.check_call([CLANG, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                           "-I" + os.path.join(ROOT, "include"), "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "plan_driver.cpp"), os.path.join(CSRC, "rf_config.cpp"),
                           os.path.join(CSRC, "rf_plan.cpp"), os.path.join(CSRC, "rf_user.cpp"), os.path.join(CSRC, "rf_glsl.cpp"), os.path.join(CSRC, "rf_abi.cpp"),
                           "-L" + os.path.join(ROOT, "reforge_amd"), "-lrfhip", "-Wl,-rpath," + os.path.join(ROOT, "reforge_amd")])
    texts = [util.random_dag(np.random.RandomState(s)) for s in range(1000)]
    texts += [util.random_graph(np.random.RandomState(10000 + s)) for s in range(1000)]
    vocab = ["input", "output", "aa", "bb", "cc", "a", "x1", "gaussian5", "sharpen", "colour_grade", "->", "->", "->", ":", "{", "}", "{}", ",",
             "sigma", "1.5", "2", "-0.5", "-3", "true", "1e3", "// note\n", "/* c */", "\n", " ", "\t", "image", "_x", "a-b", "-", ">", "*/",
             "/*", "é", "0", ".5", "5.", "combination", "input_image0", "input_image1", "mix"]
    rng = np.random.RandomState(3)
    for i in range(2500):
        t = "".join(vocab[rng.randint(len(vocab))] + ("" if rng.rand() < 0.3 else " ") for _ in range(rng.randint(1, 30)))
        texts.append("input -> aa -> bb -> output\n" + t if i % 2 else t)
    blob = tmp_path / "texts.bin"
    blob.write_bytes("\x01".join(texts).encode("utf-8"))
    # stage files (rf_user.cpp: the parser that "reflects" a user type): the shipped ones mutated token by token, and token soup
    stage_dir = tmp_path / "stages"
    stage_dir.mkdir()
    import glob
    import re
    shipped = [open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "shaders", "*.stage.hip")))]
    soup = ["struct", "Params", "{", "}", ";", "float", "int", "bool", "double", "amount", "RADIUS", "=", "0", "1", "2", "static", "constexpr", "RF_INPUTS", "RF_OUTPUTS",
            "RF_BUFFER_IN", "RF_BUFFER_OUT", "(", ")", ",", "aa_image", "ToneCurve", "256", "99999", "-1", "apply", "fill", "RF_STAGE", "//", "/*", "*/", "\n", " ", "é"]
    n_stage = 700
    for k in range(n_stage):
        if k % 7 == 6:
            text = " ".join(soup[rng.randint(len(soup))] for _ in range(rng.randint(1, 60)))
        else:
            toks = re.findall(r"\w+|\s+|[^\w\s]", shipped[k % len(shipped)])
            for _ in range(rng.randint(1, 6)):
                i = rng.randint(len(toks))
                op = rng.randint(4)
                if op == 0:
                    del toks[i]
                elif op == 1:
                    toks.insert(i, toks[rng.randint(len(toks))])
                elif op == 2:
                    toks[i] = soup[rng.randint(len(soup))]
                else:
                    j = rng.randint(len(toks))
                    toks[i], toks[j] = toks[j], toks[i]
            text = "".join(toks)
        (stage_dir / ("f%04d.stage.hip" % k)).write_text(text)
    # GLSL files (rf_glsl.cpp: lexer, top-level parser, reflection, rewrites): the shipped shaders mutated token by token, and token soup
    glsl = [open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "shaders", "*.comp")))]
    gsoup = ["layout", "(", ")", "binding", "=", "0", "1", "uniform", "buffer", "readonly", "writeonly", "image2D", "sampler2D", "rgba32f", "std430", "{", "}", ";", ",", "float", "int", "uint",
             "bool", "vec4", "ivec2", "mat3", "[", "]", "[]", "void", "main", "in", "out", "inout", "const", "shared", "struct", "#define N 3\n", "#pragma rf radius 2\n", "#if 1\n", "#endif\n",
             "#version 450\n", "local_size_x", "16", "1.0", "1e", ".5f", "0x", "imageLoad", "imageStore", "gl_GlobalInvocationID", ".", "xy", "stpq", "float[](", "//", "/*", "*/", "\\\n", "\n", " ", "é", "precision", "highp"]
    n_glsl = 900
    for k in range(n_glsl):
        if k % 6 == 5:
            text = " ".join(gsoup[rng.randint(len(gsoup))] for _ in range(rng.randint(1, 80)))
        else:
            toks = re.findall(r"\w+|\s+|[^\w\s]", glsl[k % len(glsl)])
            for _ in range(rng.randint(1, 5)):
                i = rng.randint(len(toks))
                op = rng.randint(4)
                if op == 0:
                    del toks[i]
                elif op == 1:
                    toks.insert(i, toks[rng.randint(len(toks))])
                elif op == 2:
                    toks[i] = gsoup[rng.randint(len(gsoup))]
                else:
                    j = rng.randint(len(toks))
                    toks[i], toks[j] = toks[j], toks[i]
            text = "".join(toks)
        (stage_dir / ("g%04d.comp" % k)).write_text(text)
    r = subprocess.run([exe, str(blob), str(stage_dir), str(n_stage), str(n_glsl)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "texts 4500" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    m = re.search(r"stages (\d+) parsed (\d+)", r.stdout)
    assert m and int(m.group(1)) == n_stage and 20 <= int(m.group(2)) < n_stage, r.stdout      # some mutants still parse, many are refused
    m = re.search(r"glsl (\d+) translated (\d+) planned (\d+)", r.stdout)
    assert m and int(m.group(1)) == n_glsl and 50 <= int(m.group(2)) < n_glsl and int(m.group(3)) >= 20, r.stdout
