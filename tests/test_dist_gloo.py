"""CPU, world_size 2 and 3 over gloo: the row-strip partition and the ghost-row
schedules the product computes (rf_strip_rows, rf_plan_launch_*, rf_plan_halo_schedule --
host-only C-ABI calls) are replayed with the CPU oracle as the per-strip kernel and
torch.distributed (gloo) as the neighbour exchange, and must reproduce the full-frame
result bit for bit.  This is the N>1 path of rf_graph.cpp (exchange_rows / launch_geom)
with RCCL swapped for gloo and the HIP kernels swapped for the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import reforge_amd as rf  # noqa: E402
from oracle import graph as og  # noqa: E402
from oracle import pixel  # noqa: E402
from tests import util  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _member_chain_text(cfg, members, slots=None):
    slots = slots or [0] * len(members)
    if any(slots):          # a fused fork/join launch: [nodes before the fork] [branch 0] [branch 1] [the join] [nodes after it]
        first = min(i for i, s in enumerate(slots) if s)
        last = max(i for i, s in enumerate(slots) if s)
        pre, a = members[:first], [m for m, s in zip(members, slots) if s == 1]
        b, mx, post = [m for m, s in zip(members, slots) if s == 2], members[last + 1], members[last + 2:]
        src = pre[-1] if pre else "input"
        lines = []
        if pre:
            lines.append("input -> " + " -> ".join(pre))
        lines += [" -> ".join([src] + a + [mx + ":input_image0"]), " -> ".join([src] + b + [mx + ":input_image1"]), " -> ".join([mx] + post + ["output"])]
    else:
        lines = ["input -> " + " -> ".join(members) + " -> output"]
    for m in members:
        t = cfg.type_of(m)
        params = cfg.params_of(m)
        body = "{ " + ", ".join("%s: %s" % kv for kv in sorted(params.items())) + " }" if params else "{}"
        lines.append("%s: %s %s" % (m, t, body))
    return "\n".join(lines)


def _exchange(arr, ghost, Hs, r, rank, world):
    """exchange_rows of rf_graph.cpp: top r rows -> rank-1's bottom ghost, bottom r rows
    -> rank+1's top ghost, and the symmetric receives."""
    t = torch.from_numpy(arr)
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, t[ghost:ghost + r].contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, t[ghost - r:ghost], rank - 1))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, t[ghost + Hs - r:ghost + Hs].contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, t[ghost + Hs:ghost + Hs + r], rank + 1))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()


def _worker(rank, world, port, text, W, H, fmt, exchange, fused, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flags = 0 if fused else rf.RF_GRAPH_NO_FUSION
        util.register_user_types()              # shaders/*.stage.hip on both sides (the library's lookup, the oracle's host build)
        plan = rf.Plan(rf.Config(text), flags)
        launches = plan.launch_info()
        need_src, need_dst, need_input, ghost = plan.halo_schedule(exchange)
        y0, y1 = rf.strip_rows(H, world, rank)
        Hs = y1 - y0
        assert ghost <= H // world
        cfg = og.parse_config(text)
        dtype = pixel.dtype_of(fmt)
        images = {name: np.full((Hs + 2 * ghost, W, 4), 77, dtype) for name in plan.images()}   # ghost rows start as garbage

        # upload: every rank holds ONLY its own rows of the frame
        frame = pixel.fill_synthetic(W, H, fmt, 0x5EED0004)
        images[rf.FILE_INPUT][ghost:ghost + Hs] = frame[y0:y1]
        if not exchange and need_input > 0:
            _exchange(images[rf.FILE_INPUT], ghost, Hs, need_input, rank, world)      # after_input_write

        for i, L in enumerate(launches):
            if exchange and L["radius"] > 0:
                for s in L["inputs"]:
                    _exchange(images[s], ghost, Hs, L["radius"], rank, world)          # run_launch
            # launch_geom
            row_lo, row_hi = max(-need_src[i], -y0), min(Hs - 1 + need_src[i], H - 1 - y0)
            o0, o1 = max(-need_dst[i], -y0), min(Hs + need_dst[i], H - y0)
            srcs = [images[s][ghost + row_lo:ghost + row_hi + 1] for s in L["inputs"]]
            if cfg.type_of(L["members"][0]) == "split_luma" and len(L["members"]) == 1:
                # a node with two output images: both get the rows (incl. the ghost rows) their readers want
                src = np.ascontiguousarray(srcs[0])
                luma, chroma = pixel.split_luma(src, np.empty_like(src), np.empty_like(src))
                by_name = {plan.resolve(L["members"][0] + ":luma_image"): luma, plan.resolve(L["members"][0] + ":chroma_image"): chroma}
                for name in L["outputs"]:
                    images[name][ghost + o0:ghost + o1] = by_name[name][o0 - row_lo:o1 - row_lo]
                continue
            ut = og.NODE_TYPES.get(cfg.type_of(L["members"][0]), {}).get("user")
            if ut is not None and ut.multi and len(L["members"]) == 1:
                # a user NODE (a stage file that declares its images): several inputs, one image per wired output binding
                from oracle import user_stage
                node = L["members"][0]
                srcs_c = [np.ascontiguousarray(a) for a in srcs]
                outs = [np.empty_like(srcs_c[0]) for _ in ut.outputs]
                params = og.synthesize(cfg)[node].params
                user_stage.run(ut, params, srcs_c, outs)
                by_name = {plan.resolve("%s:%s" % (node, nm)): o for nm, o in zip(ut.outputs, outs)}
                for name in L["outputs"]:
                    images[name][ghost + o0:ghost + o1] = by_name[name][o0 - row_lo:o1 - row_lo]
                continue
            if cfg.type_of(L["members"][0]) == "combination" and len(L["members"]) == 1:
                t = float(np.float32(float(cfg.params_of(L["members"][0])["mix"])))
                res = pixel.mix(srcs[0], srcs[1], t)
            else:
                res = util.run_oracle(_member_chain_text(cfg, L["members"], L["member_slots"]), np.ascontiguousarray(srcs[0]))
            images[L["output"]][ghost + o0:ghost + o1] = res[o0 - row_lo:o1 - row_lo]

        out = images[plan.resolve(rf.FINAL_OUTPUT)][ghost:ghost + Hs]
        np.save(os.path.join(out_dir, "strip%d.npy" % rank), out)
    finally:
        dist.destroy_process_group()


CASES = [
    # text, world, exchange, fused
    (util.CHAIN5, 2, True, False),
    (util.CHAIN5, 2, True, True),
    (util.CHAIN5, 2, False, True),
    (util.CHAIN5, 3, False, False),
    (util.DIAMOND, 2, True, True),
    (util.CHAIN3, 3, True, True),
    (util.SPLIT2, 2, True, False),       # a node with two output images, exchange and over-fetch
    (util.SPLIT2, 3, False, True),
]
# user types: a node with two inputs and two outputs read (its own kernel), row stages fused with a gaussian
USER_BOTH = """
input -> blur -> um:blurred_image
input -> um:input_image
um -> mm:input_image0
um:mask_image -> ee -> mm:input_image1
mm -> nn -> output
blur: gaussian5 { sigma: 1.0 }
um: unsharp_mask { amount: 0.8, threshold: 0.05 }
ee: edge_detect { scale: 1.5 }
mm: combination { mix: 0.25 }
nn: invert { enabled: true, strength: 0.6 }
"""
CASES += [(USER_BOTH, 2, True, True), (USER_BOTH, 3, False, False), (USER_BOTH, 2, False, True)]
# graphs nobody wrote by hand (tests/util.py::random_graph), exchange and over-fetch schedules
for _seed, _world, _xchg, _fused in ((3001, 2, True, True), (3002, 3, True, False), (3003, 2, False, True), (3004, 2, True, True)):
    CASES.append((util.random_graph(np.random.RandomState(_seed)), _world, _xchg, _fused))


@pytest.mark.parametrize("text,world,exchange,fused", CASES)
@pytest.mark.parametrize("fmt", [util.F32, util.U8])
def test_row_strips_reproduce_the_full_frame(tmp_path, text, world, exchange, fused, fmt):
    W, H = 29, 41
    port = _free_port()
    mp.spawn(_worker, args=(world, port, text, W, H, fmt, exchange, fused, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / ("strip%d.npy" % r)) for r in range(world)], axis=0)
    old = util.register_user_types()
    try:
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, 0x5EED0004))
    finally:
        rf.set_shader_path(old)
    util.assert_same(got, want, "row strips, world=%d exchange=%s fused=%s" % (world, exchange, fused))


def test_halo_schedules_kat():
    """Exchange mode: each launch exchanges its own radius.  Over-fetch: the input
    carries the cumulative halo (2+1+4 = 7 rows for the 5-stage chain)."""
    p = rf.Plan(rf.Config(util.CHAIN5), rf.RF_GRAPH_NO_FUSION)
    assert [l["radius"] for l in p.launch_info()] == [2, 0, 1, 4, 0]
    assert p.halo_schedule(True) == ([2, 0, 1, 4, 0], [0, 0, 0, 0, 0], 0, 4)
    assert p.halo_schedule(False) == ([7, 5, 5, 4, 0], [5, 5, 4, 0, 0], 7, 7)
    p = rf.Plan(rf.Config(util.CHAIN5), 0)                      # the whole chain is one launch
    assert [l["radius"] for l in p.launch_info()] == [7]
    assert p.halo_schedule(False) == ([7], [0], 7, 7)
    two = util.CHAIN5.replace("gaussian9    { sigma: 2.0 }", "gaussian5    { sigma: 2.0 }")   # catalogue-only fusion splits it 3 + 2
    p = rf.Plan(rf.Config(two), rf.RF_GRAPH_NO_JIT)
    assert [l["radius"] for l in p.launch_info()] == [3, 2]
    assert p.halo_schedule(False) == ([5, 2], [2, 0], 5, 5)
