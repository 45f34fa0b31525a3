"""CPU: the ISA invariants of tests/test_isa_invariants.py over kernels the RUN-TIME compiler produces (hiprtc, rf_jit.cpp).

The ahead-of-time catalogue is checked on hipcc's assembly listing; the open-ended instantiations a graph can ask for at
rf_graph_create were not looked at by anything -- and that is where round 2's store-data hazard hid (a run-time compiled
fork/join kernel scheduled a VALU write one wait state behind the inline-asm store).  Here generated chains and fork/joins
are compiled through the C ABI (rf_plan_jit_compile: no device needed) into a disk cache, and the CODE OBJECTS are
disassembled (llvm-objdump) and checked: per steady-loop row T DMAs and T stores and nothing else on the vector-memory
path, DMA -> counted wait -> store order, an `s_nop 1` behind every global_store_dwordx4, `lgkmcnt(0)` in front of every
DMA refill, no scratch, no barrier."""
import glob
import os
import re
import sys

import pytest

import reforge_amd as rf
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_obj  # noqa: E402
import isa_util  # noqa: E402

pytestmark = pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")

from tests import jit_isa_compile as gen  # noqa: E402


def _torch_importable():
    import importlib.util
    return importlib.util.find_spec("torch") is not None


@pytest.fixture(scope="module", params=["this process", "a process that imported torch first"])
def compiled(request, tmp_path_factory):
    """compile ~50 generated launches for both formats into a private cache dir; returns the .hsaco paths.
    Twice: in this process (whatever libhiprtc it has), and in a child that imports PyTorch FIRST and therefore compiles with the
    libhiprtc / libamd_comgr PyTorch bundles -- another compiler build, which gave the sharpen parameters a scratch copy until
    they were stored as aligned pairs (StCross3::Params)."""
    cache = tmp_path_factory.mktemp("jitcache")
    if request.param == "this process":
        old = os.environ.get("RF_JIT_CACHE_DIR")
        os.environ["RF_JIT_CACHE_DIR"] = str(cache)
        try:
            n_jit = gen.compile_all(rf, os.path.join(ROOT, "shaders"))
        finally:
            if old is None:
                os.environ.pop("RF_JIT_CACHE_DIR", None)
            else:
                os.environ["RF_JIT_CACHE_DIR"] = old
        assert n_jit >= 20
    else:
        if not _torch_importable():
            pytest.skip("no PyTorch here")
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "jit_isa_compile.py"), str(cache), "torch"], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and "compiled" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(glob.glob(str(cache / "*.hsaco")))
    assert len(files) >= 24, len(files)
    return files


def pf_t(name):
    m = re.search(r"stream_kernelINS_\w+?ELi(\d+)ELi(\d+)E", name)
    return int(m.group(1)), int(m.group(2))


def test_cache_entries_are_consistent(compiled):
    for f in compiled:
        name_file = f[:-6] + ".name"
        lines = open(name_file).read().split("\n")
        assert lines[0].startswith("_ZN2rf13stream_kernel"), name_file
        size, _hash = lines[1].split()
        assert int(size) == os.path.getsize(f)
        assert not glob.glob(f[:-6] + "*.tmp*")


def test_run_time_compiled_kernels_keep_the_counted_wait_contract(compiled):
    steady_checked = 0
    for f in compiled:
        funcs = isa_obj.functions(f)
        ks = {n: ins for n, ins in funcs.items() if n.startswith("_ZN2rf13stream_kernel")}
        assert len(ks) == 1, (f, list(funcs))
        (name, ins), = ks.items()
        ops = [i.op for i in ins]
        assert "s_barrier" not in ops, name
        assert not any(o.startswith("scratch_") for o in ops), name                    # no spills
        assert not any(o.startswith(("global_load_dword", "buffer_load", "flat_")) and not o.startswith("global_load_lds") for o in ops), name
        bad = isa_util.scalar_base_violations([i.text for i in ins])      # RF_SBASE, rf_device.h
        assert not bad, (name, bad[:3])
        for k, i in enumerate(ins):
            if i.op.startswith("global_load_lds"):
                assert any(p.op == "s_waitcnt" and "lgkmcnt(0)" in p.text for p in ins[max(0, k - 5):k]), (name, i.text)
            if i.op == "global_store_dwordx4":
                assert ins[k + 1].text.split()[:2] == ["s_nop", "1"], (name, ins[k + 1].text)
        pf, t = pf_t(name)
        steady, warm = "vmcnt(%d)" % ((2 * pf - 2) * t), "vmcnt(%d)" % ((pf - 1) * t)
        found = 0
        for _simple, seq in isa_obj.loops(ins):
            text = "\n".join(i.text for i in seq)
            if "s_waitcnt " + steady not in text or warm in text or "vmcnt(0)" in text:
                continue
            lops = [i.op for i in seq]
            vmem = sorted(o for o in lops if o.startswith(("global_", "buffer_", "flat_")))
            assert vmem in (["global_load_lds_dword"] * t + ["global_store_dword"] * t,
                            ["global_load_lds_dwordx4"] * t + ["global_store_dwordx4"] * t), (name, vmem)
            i_dma = max(k for k, o in enumerate(lops) if o.startswith("global_load_lds"))
            i_wait = next(k for k, i in enumerate(seq) if steady in i.text)
            i_store = min(k for k, o in enumerate(lops) if o.startswith("global_store"))
            assert i_dma < i_wait < i_store, name
            found += 1
        assert found >= 1, name
        steady_checked += found
    assert steady_checked >= len(compiled)


def test_two_texel_variants_keep_the_contract_or_are_rejected(tmp_path, monkeypatch):
    """the two-texels-per-lane variant of a run-time compiled chain is built under the same 256-VGPR bound with twice
    the loop-carried state.  rf_graph_create drops it when it spills (stream_prepare -> jit_forget); a variant that does
    NOT spill must keep the 2-DMA / 2-store contract its counted waits (vmcnt(12) in the steady loop) rest on."""
    monkeypatch.setenv("RF_JIT_CACHE_DIR", str(tmp_path))
    texts = ["input -> aa -> bb -> cc -> output\naa: gaussian { sigma: 1.0, radius: 3 }\nbb: colour_grade { slope: 1.0, offset: 0.0, saturation: 1.0 }\ncc: gaussian5 { sigma: 1.0 }",
             "input -> aa -> bb -> cc -> dd -> output\naa: sharpen { amount: 0.3 }\nbb: gaussian9 { sigma: 2.0 }\ncc: sharpen { amount: 0.2 }\ndd: gaussian { sigma: 1.0, radius: 1 }",
             "input -> aa -> bb -> output\naa: gaussian { sigma: 1.0, radius: 3 }\nbb: gaussian { sigma: 1.0, radius: 4 }"]
    for text in texts:
        p = rf.Plan(rf.Config(text))
        assert len(p.launches()) == 1 and p.needs_jit() == [True], text
        assert p.jit_compile(rf.RF_FORMAT_RGBA32F, texels_per_lane=2) > 4096
    files = sorted(glob.glob(str(tmp_path / "*.hsaco")))
    assert len(files) == len(texts)
    kept = 0
    for f in files:
        (name, ins), = [(n, v) for n, v in isa_obj.functions(f).items() if n.startswith("_ZN2rf13stream_kernel")]
        pf, t = pf_t(name)
        assert t == 2
        if any(i.op.startswith("scratch_") for i in ins):
            continue                               # the product refuses this one (scratch_bytes > 0): nothing more to hold it to
        kept += 1
        steady = "vmcnt(%d)" % ((2 * pf - 2) * t)
        found = 0
        for _simple, seq in isa_obj.loops(ins):
            text = "\n".join(i.text for i in seq)
            if "s_waitcnt " + steady not in text or "vmcnt(%d)" % ((pf - 1) * t) in text or "vmcnt(0)" in text:
                continue
            lops = [i.op for i in seq]
            assert sorted(o for o in lops if o.startswith(("global_", "buffer_", "flat_"))) == ["global_load_lds_dwordx4"] * 2 + ["global_store_dwordx4"] * 2, name
            found += 1
        assert found >= 1, name
    assert kept >= 1
