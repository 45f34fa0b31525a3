"""GPU: the DYNAMIC TAIL of the stream launches (rf_stream_dev.h, "Walk words").

A wave that has finished its rows takes over the far half of the longest unfinished walk it finds; which wave writes a row
is all that changes, so every result must stay bit-identical to the oracle -- for every stage kind, both formats, both walk
directions, one and two texels per lane, launches over two row ranges, in-place nodes (where a row written twice would show)
and frame after frame on the same graph (the walk words must be empty again when a launch ends).  The launches here are
forced dynamic (RF_EXEC_DYNAMIC_WALKS) with chunks of unequal height, so that walks really are taken over: the library
counts them (rf_graph_walks_taken) and the tests require the count to move."""
import numpy as np
import pytest

import reforge_amd as rf
from oracle import pixel
from tests import util
from tests.test_gpu_parity import NODES

pytestmark = pytest.mark.gpu
DYN = rf.RF_EXEC_DYNAMIC_WALKS


def run(ctx, text, img, frames=1, **kw):
    """-> (output of the last frame, walks taken over)"""
    H, W, _ = img.shape
    g = rf.Graph(ctx, rf.Config(text), W, H, pixel.fmt_of(img), **kw)
    try:
        out = None
        for _ in range(frames):
            g.upload_raw(img)
            g.execute()
            g.wait()
            out = g.download_raw()
        return out, g.walks_taken()
    finally:
        g.close()


@pytest.mark.parametrize("fmt", [util.F32, util.U8])
@pytest.mark.parametrize("name", ["passthrough", "gaussian5", "gaussian9", "gaussian_r7", "gaussian_r15", "grade", "sharpen", "chain3", "chain5",
                                  "chain5_split", "diamond", "inplace"])
def test_dynamic_walks_match_the_oracle(ctx, name, fmt):
    W, H = 333, 700
    x = util.synthetic(W, H, fmt, seed=0xD1A0 + len(name))
    want = util.run_oracle(NODES[name], x)
    taken = 0
    # chunks of 260 + 260 + 180 rows: the waves of the short chunk finish first and take over parts of the tall ones;
    # one chunk per strip: only the idle waves of the last strip group have nothing to do -- they start by taking walks over
    for rpc, unit, ex in ((260, 8, DYN), (260, 12, DYN | rf.RF_EXEC_ALTERNATE), (260, 8, DYN | rf.RF_EXEC_NO_ALTERNATE), (100000, 16, DYN)):
        got, n = run(ctx, NODES[name], x, rows_per_chunk=rpc, walk_unit=unit, exec_flags=ex)
        util.assert_same(got, want, "%s rpc=%d unit=%d flags=%#x" % (name, rpc, unit, ex))
        taken += n
    assert taken > 0, "no walk was ever taken over: the schedule under test was the static one"
    # the same launches with the static schedule, and unfused
    got, n = run(ctx, NODES[name], x, rows_per_chunk=260, exec_flags=rf.RF_EXEC_STATIC_WALKS)
    util.assert_same(got, want, name + " static")
    assert n == 0
    got, _ = run(ctx, NODES[name], x, rows_per_chunk=260, walk_unit=8, exec_flags=DYN, flags=rf.RF_GRAPH_NO_FUSION)
    util.assert_same(got, want, name + " unfused")


SIZES = [(1, 1), (2, 3), (5, 40), (1, 90), (17, 13), (61, 64), (121, 135), (250, 331), (64, 257), (300, 97)]


@pytest.mark.parametrize("name", ["gaussian9", "gaussian_r15", "chain3", "chain5", "sharpen", "inplace"])
def test_dynamic_walks_on_ragged_and_tiny_frames(ctx, name):
    """frames of 1 x 1 ... 300 x 97: grids of one to a few workgroups, chunks shorter than a unit, walks of one unit, chunks whose
    last unit is all there is -- every geometry in which a word is published, or not, at the edge of the rules"""
    for fmt in (util.F32, util.U8):
        for W, H in SIZES:
            x = util.synthetic(W, H, fmt, seed=0x5EED0000 + W * 131 + H)
            want = util.run_oracle(NODES[name], x)
            for rpc in (0, 24, 64, 100000):
                got, _ = run(ctx, NODES[name], x, rows_per_chunk=rpc, walk_unit=8, exec_flags=DYN)
                util.assert_same(got, want, "%s %dx%d rpc=%d" % (name, W, H, rpc))


def test_dynamic_walks_with_two_texels_per_lane(ctx):
    W, H = 700, 640
    x = util.synthetic(W, H, util.F32, seed=77)
    taken = 0
    for name in ("gaussian9", "chain3", "chain5", "passthrough"):
        want = util.run_oracle(NODES[name], x)
        for rpc in (250, 100000):
            got, n = run(ctx, NODES[name], x, rows_per_chunk=rpc, walk_unit=8, texels_per_lane=2, exec_flags=DYN)
            util.assert_same(got, want, "%s two texels rpc=%d" % (name, rpc))
            taken += n
    assert taken > 0


def test_dynamic_walks_frame_after_frame(ctx):
    """the walk words of a launch are empty again when it ends: the next frame on the same graph starts from them"""
    W, H = 333, 700
    for fmt in (util.F32, util.U8):
        x = util.synthetic(W, H, fmt, seed=5)
        for name in ("chain3", "inplace", "gaussian9"):
            want = util.run_oracle(NODES[name], x)
            got, n = run(ctx, NODES[name], x, frames=6, rows_per_chunk=260, walk_unit=8, exec_flags=DYN)
            util.assert_same(got, want, name + " sixth frame")
            assert n > 0


def test_dynamic_walks_in_a_split_launch_and_on_several_slots(ctx):
    """the interior of a split launch runs dynamic, its boundary slivers (two row ranges in one launch) static; frame slots have
    walk words of their own"""
    W, H = 333, 700
    x = util.synthetic(W, H, util.F32, seed=9)
    for name in ("chain3", "gaussian9", "chain5"):
        want = util.run_oracle(NODES[name], x)
        got, _ = run(ctx, NODES[name], x, rows_per_chunk=260, walk_unit=8, exec_flags=DYN | rf.RF_EXEC_FORCE_SPLIT)
        util.assert_same(got, want, name + " split")
        g = rf.Graph(ctx, rf.Config(NODES[name]), W, H, util.F32, num_frames=3, rows_per_chunk=260, walk_unit=8, exec_flags=DYN)
        try:
            g.upload_raw(x)              # every slot's input image
            for s in range(3):
                g.execute(s)
            for s in range(3):
                g.wait(s)
                util.assert_same(g.download_raw(s), want, "%s slot %d" % (name, s))
        finally:
            g.close()


def test_the_library_chooses_dynamic_walks_for_a_large_launch(ctx):
    """no flags: a launch that fills the chip takes tall chunks and walk words by itself (and still equals the static schedule)"""
    W, H = 3840, 2160
    g = rf.Graph(ctx, rf.Config(NODES["gaussian9"]), W, H, util.F32)
    s = rf.Graph(ctx, rf.Config(NODES["gaussian9"]), W, H, util.F32, exec_flags=rf.RF_EXEC_STATIC_WALKS)
    try:
        for k in (g, s):
            k.fill_synthetic(0x5EED0003)
            for _ in range(3):
                k.execute()
            k.wait()
        a, b = g.download_raw(), s.download_raw()
        assert a.tobytes() == b.tobytes()
        assert s.walks_taken() == 0
    finally:
        g.close()
        s.close()
