"""CPU: the host-only half of the library (config parser, planner, ABI accessors) built with
AddressSanitizer + UBSan + LeakSanitizer and fed 4500 generated configs (valid graphs from both
generators and token soup).  GPU sanitizers are not available on the pool; this covers the code
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
                           os.path.join(CSRC, "rf_plan.cpp"), os.path.join(CSRC, "rf_abi.cpp"),
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
    r = subprocess.run([exe, str(blob)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "texts 4500" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
