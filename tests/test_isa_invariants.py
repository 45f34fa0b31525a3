"""CPU: invariants of the generated gfx950 code that the stream kernel's correctness rests on.

The kernel waits for its LDS-DMA rows with a COUNTED `s_waitcnt vmcnt(N)` (rf_stream.hip,
Source::wait_row): that is only right if, per row, a wave issues exactly T vector-memory loads
(the DMAs; T = texels per lane) and T vector-memory stores, in a fixed order.  hipcc
cross-compiles without a GPU, so the assembly is checked here: every steady loop of every
stream_kernel instantiation has exactly T `global_load_lds_*`, exactly T `global_store_*`, no
other vector-memory instruction, and the kernels use no scratch and no workgroup barrier."""
import os
import re
import subprocess

import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from isa_util import innermost_loops, scalar_base_violations  # noqa: E402
CSRC = os.path.join(ROOT, "reforge_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def stream_asm(tmp_path_factory):
    # the listing the BUILD kept beside rf_stream.o (Makefile: -save-temps=obj) is the code that ships; it is used when it is
    # at least as new as every source it depends on, otherwise the file is compiled here (3 minutes)
    kept = os.path.join(CSRC, "build", "rf_stream-hip-amdgcn-amd-amdhsa-gfx950.s")
    deps = [os.path.join(CSRC, f) for f in ("rf_stream.hip", "rf_stream_dev.h", "rf_device.h", "rf_kernels.h", "rf_jit.h", "rf_user.h", "rf_plan.h")]
    if os.path.exists(kept) and os.path.getmtime(kept) >= max(os.path.getmtime(d) for d in deps):
        return open(kept).read().split("\n")
    out = tmp_path_factory.mktemp("isa") / "rf_stream.s"
    subprocess.check_call([HIPCC, "-S", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(CSRC, "rf_stream.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


def kernels(lines):
    """name -> list of instruction lines (labels kept) for every stream_kernel function"""
    out, name, body = {}, None, []
    for l in lines:
        m = re.match(r"^(_ZN2rf13stream_kernel\w+):", l)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(l)
            if "s_endpgm" in l:
                out[name] = body
                name = None
    return out


def prefetch_depth(name):
    """(PF, T) of stream_kernel<Px, PF, T, stages...>"""
    m = re.search(r"stream_kernelINS_\w+?ELi(\d+)ELi(\d+)E", name)
    return int(m.group(1)), int(m.group(2))


def steady_loops(body, pf=4, t=1):
    """instruction lists of the loops whose wait is the steady counted form vmcnt((2*PF-2)*T)"""
    steady, warm = "vmcnt(%d)" % ((2 * pf - 2) * t), "vmcnt(%d)" % ((pf - 1) * t)
    loops = []
    for _, ins in innermost_loops(body):      # by basic-block membership: hipcc may lay a loop's latch out before its header
        text = "\n".join(ins)
        if "s_waitcnt " + steady not in text:
            continue
        if warm in text or "vmcnt(0)" in text:
            continue                          # the generic (priming / tail) loop carries all three waits
        loops.append(ins)
    return loops


def test_stream_kernels_keep_the_counted_wait_contract(stream_asm):
    ks = kernels(stream_asm)
    assert len(ks) >= 40                      # every node type / fused pattern / format / prefetch depth
    assert sum("7PxF32NT" in n for n in ks) >= 20
    checked = 0
    for name, body in ks.items():
        text = "\n".join(body)
        assert "s_barrier" not in text, name                    # wave-autonomous: no workgroup barrier
        assert "scratch_" not in text, name                     # no spills
        assert re.search(r"global_load_lds_dword(x4)?\b", text), name
        assert not re.search(r"\bglobal_load_dword", text.replace("global_load_lds_dword", "")), name   # the DMA is the only load
        # every DMA refill is issued behind an lgkmcnt(0): the ds_read that consumed the slot's previous
        # row has executed (an L2-hit refill could otherwise overtake it; Source::issue)
        instrs = [x.strip() for x in body if x.strip() and not x.strip().startswith((";", "."))]
        for i, ins in enumerate(instrs):
            if ins.startswith("global_load_lds"):
                assert any(p.startswith("s_waitcnt") and "lgkmcnt(0)" in p for p in instrs[max(0, i - 5):i]), name
        # every vector-memory instruction takes its scalar base from an s_mov_b64 of its own asm statement (a base reloaded by
        # v_readlane right in front of the asm would be read stale: the round-3 / round-4 memory faults of the 27- and 31-tap gaussians)
        assert not scalar_base_violations(instrs), (name, scalar_base_violations(instrs)[:3])
        # the rgba32f store is inline asm, so hipcc's hazard recogniser cannot keep a VALU write to its data registers two
        # wait states away (gfx940+: stores of more than 8 bytes): the asm carries its own s_nop 1
        for i, ins in enumerate(instrs):
            if ins.startswith("global_store_dwordx4"):
                assert instrs[i + 1].split()[:2] == ["s_nop", "1"], (name, instrs[i:i + 3])
        # the non-temporal hint is carried by every row store of the PxF32NT variants (launches whose result no launch reads)
        # and by nothing else -- not by rgba8 stores, not by any row load (halo rows are re-used through L2)
        stores = [x for x in instrs if x.startswith("global_store_dword")]
        if "7PxF32NT" in name:
            assert stores and all(x.split()[-1] == "nt" for x in stores), (name, stores[:2])
        else:
            assert not any(" nt" in x for x in stores), (name, stores[:2])
        assert not any(" nt" in x for x in instrs if x.startswith("global_load_lds")), name
        pf, t = prefetch_depth(name)
        for loop in steady_loops(body, pf, t):
            ops = [x.split()[0] for x in loop]
            vmem = [o for o in ops if o.startswith(("global_", "buffer_", "flat_"))]
            assert sorted(vmem) in (["global_load_lds_dword"] * t + ["global_store_dword"] * t,
                                    ["global_load_lds_dwordx4"] * t + ["global_store_dwordx4"] * t), (name, vmem)
            # program order inside an iteration: the T DMAs, then the counted wait, then the T stores
            i_dma = max(k for k, o in enumerate(ops) if o.startswith("global_load_lds"))
            i_wait = next(k for k, x in enumerate(loop) if "vmcnt(%d)" % ((2 * pf - 2) * t) in x)
            i_store = min(k for k, o in enumerate(ops) if o.startswith("global_store"))
            assert i_dma < i_wait < i_store, name
            checked += 1
    assert checked >= len(ks)                 # at least one steady loop each (two when the pipeline has a halo)


def test_fused_chain_steady_loop_is_lean(stream_asm):
    """The 3-stage rgba32f chain: all 68 fmaf per row are packed (v_pk_fma_f32), and the loop
    stays near 110 instructions (it was ~250 before the steady-state split)."""
    ks = kernels(stream_asm)
    name = [n for n in ks if "PxF32ELi4ELi1E" in n and "StHTapILi2E" in n and "StGrade" in n and "StCross3EEEEv" in n]   # ends at the sharpen
    assert len(name) == 1
    loops = steady_loops(ks[name[0]])
    assert len(loops) == 2                    # top-down (vertical taps in scatter form) and bottom-up walks
    for loop in loops:
        ops = [x.split()[0] for x in loop]
        assert ops.count("v_pk_fma_f32") >= 30
        assert len(ops) <= 130, len(ops)


@pytest.fixture(scope="module")
def conv_asm(tmp_path_factory):
    kept = os.path.join(CSRC, "build", "rf_conv-hip-amdgcn-amd-amdhsa-gfx950.s")
    deps = [os.path.join(CSRC, f) for f in ("rf_conv.hip", "rf_device.h", "rf_kernels.h")]
    if os.path.exists(kept) and os.path.getmtime(kept) >= max(os.path.getmtime(d) for d in deps):
        return open(kept).read().split("\n")
    out = tmp_path_factory.mktemp("isa") / "rf_conv.s"
    subprocess.check_call([HIPCC, "-S", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(CSRC, "rf_conv.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


def test_conv_valu_kernels_take_their_weights_as_scalar_operands(conv_asm):
    """rf_conv.hip, conv2d_valu_kernel: the launch is bound by the package power, and what moved it (0.67-0.69 -> 0.70-0.75 of the f32
    peak) was the weight operand of every packed FMA becoming a scalar register pair -- s_load of the weight row through the scalar
    cache, `op_sel` broadcasting it -- instead of a VGPR filled from an LDS copy.  A compiler that loses that (a vector load of a
    uniform address, or an unrolled weight-row loop that spills: hipcc did that for K <= 7 until the loop was pinned) costs 8 % to 8x
    and no parity test would notice."""
    text = "\n".join(conv_asm)
    seen = 0
    for px in ("5PxF32", "4PxU8"):
        for K in range(7, 32, 2):
            m = re.search(r"^(_ZN2rf18conv2d_valu_kernelINS_%sELi%dELi8ELi8E\w+):[^\n]*\n(.*?)s_endpgm" % (px, K), text, re.S | re.M)
            assert m, (px, K)
            body = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", "."))]
            assert not any(l.startswith("scratch_") for l in body), (px, K)
            fmas = [l for l in body if l.startswith("v_pk_fma_f32")]
            assert len(fmas) >= 2 * 8 * K and all(re.search(r", s\[\d+:\d+\]", l) for l in fmas), (px, K, len(fmas))       # K taps x 8 columns x 2 halves, each with an SGPR-pair operand
            assert any(l.startswith("s_load_dword") for l in body)
            # LDS: the register window only -- K + 7 texel reads per weight row, no weight reads
            assert sum(l.startswith("ds_read_b128") for l in body) <= K + 7 + 2, (px, K)
            seen += 1
    assert seen == 26
