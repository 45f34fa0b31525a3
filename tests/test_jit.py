"""CPU: kernels compiled at graph creation (rf_jit.cpp).  hiprtc cross-compiles gfx950 without a GPU, exactly
like hipcc does for the ahead-of-time catalogue, so the generated stream_kernel<> instantiations are built
here; the GPU parity tests run them."""
import os

import pytest

import reforge_amd as rf
from tests import util

pytestmark = pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")

MIXED = """input -> aa -> bb -> cc -> dd -> ee -> output
aa: gaussian { sigma: 1.5, radius: 3 }
bb: passthrough {}
cc: sharpen { amount: 0.4 }
dd: colour_grade { slope: 1.1, offset: 0.0, saturation: 1.0 }
ee: gaussian5 { sigma: 0.8 }
"""


def test_a_chain_outside_the_catalogue_is_one_launch_and_its_kernel_builds_for_gfx950():
    p = rf.Plan(rf.Config(MIXED))
    assert p.launches() == ["aa+bb+cc+dd+ee"] and p.needs_jit() == [True]
    for fmt in (rf.RF_FORMAT_RGBA32F, rf.RF_FORMAT_RGBA8):
        assert p.jit_compile(fmt) > 4096            # a gfx950 code object came back
    # the BASELINE chains are in the catalogue: nothing to compile
    for text in (util.CHAIN3, util.CHAIN5):
        q = rf.Plan(rf.Config(text))
        assert q.needs_jit() == [False] and q.jit_compile() == 0


def test_no_jit_flag_and_env_plan_with_the_catalogue_alone(monkeypatch):
    assert rf.Plan(rf.Config(MIXED), rf.RF_GRAPH_NO_JIT).needs_jit() == [False, False, False]
    monkeypatch.setenv("RF_NO_JIT", "1")
    assert not rf.lib().rf_jit_available()
    assert len(rf.Plan(rf.Config(MIXED)).launches()) == 3


def test_the_embedded_device_source_is_the_checked_in_headers():
    """build/rf_jit_source.inc is generated from rf_device.h + rf_stream_dev.h + rf_user_dev.h by the Makefile: the text the run-time
    compiler sees must be the text the ahead-of-time kernels were built from."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "reforge_amd", "csrc")
    inc = open(os.path.join(csrc, "build", "rf_jit_source.inc")).read()
    for f in ("rf_device.h", "rf_stream_dev.h", "rf_user_dev.h"):
        body = [l for l in open(os.path.join(csrc, f)).read().split("\n") if l.strip() and not l.startswith("#include \"") and l != "#pragma once"]
        for l in body[:: max(1, len(body) // 50)]:
            assert l in inc, (f, l)
    assert os.path.getmtime(os.path.join(csrc, "build", "rf_jit_source.inc")) >= max(os.path.getmtime(os.path.join(csrc, f)) for f in ("rf_device.h", "rf_stream_dev.h", "rf_user_dev.h"))


def test_admission_rule_keeps_oversized_chains_split():
    wide = "input -> aa -> bb -> output\naa: gaussian { sigma: 4.0, radius: 12 }\nbb: gaussian { sigma: 4.0, radius: 12 }"
    assert rf.Plan(rf.Config(wide)).launches() == ["aa", "bb"]
    nine = "input -> " + " -> ".join("n%d" % i for i in range(9)) + " -> output\n" + "\n".join("n%d: colour_grade {}" % i for i in range(9))
    assert max(len(l.split("+")) for l in rf.Plan(rf.Config(nine)).launches()) <= 8       # kMaxFusedOps


def test_generated_chains_of_fusable_nodes_are_never_split():
    """A chain of 3..8 passthrough / gaussian5 / gaussian9 / gaussian{radius <= 3} / colour_grade / sharpen nodes with
    no fork is ONE launch (the catalogue alone would cut almost every one of them into pairs), and the plan with
    run-time compiled kernels never has more launches than the catalogue-only plan."""
    import numpy as np
    rng = np.random.RandomState(4242)
    kinds = ["passthrough {}", "gaussian5 { sigma: 1.0 }", "gaussian9 { sigma: 2.0 }", "gaussian { sigma: 1.2, radius: %d }",
             "colour_grade { slope: 1.1, offset: 0.0, saturation: 0.9 }", "sharpen { amount: 0.5 }"]
    split = 0
    for trial in range(120):
        n = int(rng.randint(3, 9))
        names = ["n%02d" % i for i in range(n)]
        decl = []
        big = 0
        for nm in names:
            k = kinds[rng.randint(len(kinds))]
            if "%d" in k:
                k = k % rng.randint(0, 4)
            big += "gaussian9" in k
            decl.append("%s: %s" % (nm, k))
        text = "input -> " + " -> ".join(names) + " -> output\n" + "\n".join(decl)
        jit, cat = rf.Plan(rf.Config(text)).launches(), rf.Plan(rf.Config(text), rf.RF_GRAPH_NO_JIT).launches()
        assert len(jit) <= len(cat), text
        if n <= 5 and big <= 1:           # (long chains with several wide windows exceed the admission rule's register estimate)
            assert len(jit) == 1, (jit, text)
        split += len(cat) > 1
    assert split > 60                     # the catalogue-only plan really does split most of them


def test_generated_graphs_with_forks_and_in_place_nodes_plan_no_worse_with_jit():
    import numpy as np
    for seed in range(150):
        rng = np.random.RandomState(9000 + seed)
        text = (util.random_graph if seed % 2 else util.random_dag)(rng)
        assert len(rf.Plan(rf.Config(text)).launches()) <= len(rf.Plan(rf.Config(text), rf.RF_GRAPH_NO_JIT).launches()), text
