"""CPU: the graph oracle with USER filter types (oracle/user_stage.py compiles the stage FILES for the host).

(1) the host build of every shipped stage file reproduces the exact-rational restatements of tests/test_gpu_user_stage.py and
tests/test_gpu_user_node.py bit for bit, both formats -- so the oracle's user types are pinned by the same independent
statements the kernels are; (2) for generated graphs that mix user types with built-in nodes, the library's planner and the
restatement of order_by_execution / reusable_image_remapping agree (src/vulkan/pipeline_graph.rs:358-497)."""
import os

import numpy as np
import pytest

import reforge_amd as rf
from oracle import graph as og
from tests import util
from tests import test_gpu_user_node as gu
from tests import test_gpu_user_stage as gs
from tests.test_user_node import CURVE, GUIDED, TINT, TINT_GRAPH, UNSHARP_BOTH, WINDOW_GRAPH


@pytest.fixture
def user_types(tmp_path):
    old = util.register_user_types()
    (tmp_path / "tint.stage.hip").write_text(TINT)
    og.register_user_type("tint", str(tmp_path / "tint.stage.hip"))
    (tmp_path / "guided.stage.hip").write_text(GUIDED)
    og.register_user_type("guided", str(tmp_path / "guided.stage.hip"))
    yield
    rf.set_shader_path(old)
    og.NODE_TYPES.pop("tint", None)
    og.NODE_TYPES.pop("guided", None)


@pytest.mark.parametrize("fmt,tag", [(util.F32, "f32"), (util.U8, "u8")])
def test_host_build_of_the_stage_files_matches_the_exact_restatements(user_types, fmt, tag):
    x = util.synthetic(29, 11, fmt, 41)
    util.assert_same(util.run_oracle(UNSHARP_BOTH, x), gu.want_unsharp_both(x, tag), "unsharp_mask, two inputs, both outputs read")
    util.assert_same(util.run_oracle(TINT_GRAPH, x), gu.want_tint(x, tag), "a node that writes its input in place")
    util.assert_same(util.run_oracle(CURVE, x), gu.want_curve(x, tag, 0.6, 0.05, 0.8), "tone_curve -> apply_curve over a buffer edge")
    util.assert_same(util.run_oracle(gs.CHAIN % ("0.75", "1.0"), x), gs.want_chain(x, tag, 0.75, 1.0), "gaussian5 -> edge_detect -> invert")
    util.assert_same(util.run_oracle(WINDOW_GRAPH, x), gu.want_window_graph(x, tag), "nodes that read through windows (RADIUS 2; two inputs at RADIUS 1)")


def test_generated_graphs_with_user_types_plan_like_the_restatement(user_types):
    planned = 0
    for seed in range(150):
        text = util.random_dag(np.random.RandomState(seed), split=seed % 2 == 0, user=True)
        infos = og.synthesize(og.parse_config(text))
        layers = og.order_by_execution(infos)
        p = rf.Plan(rf.Config(text), rf.RF_GRAPH_NO_FUSION)
        assert p.layers() == layers, text
        assert p.aliases() == og.reusable_image_remapping(layers, infos), text
        assert p.images() == og.GraphOracle(text, 4, 4, util.F32).allocated_images(), text
        p.halo_schedule()                                   # the launch list exists: nothing the kernels cannot execute
        fused = rf.Plan(rf.Config(text))
        assert len(fused.launches()) <= len(p.launches())
        planned += any(t in text for t in util.USER_TYPES)
    assert planned >= 100                                   # most of the generated graphs do hold a user type
