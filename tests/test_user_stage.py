"""CPU: filter types that are FILES ({shader_path}/{type}.stage.hip, rf_user.h) -- the counterpart of the reference's
{shader_path}/{type}.comp (src/config/config.rs:59-75, src/vulkan/shader.rs:29-59,:106-160).  Parsing ("reflection"), planning,
fusion with built-in nodes, and compiling for gfx950 without a device; tests/test_gpu_user_stage.py runs them."""
import os
import shutil

import pytest

import reforge_amd as rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")

CHAIN = """input -> blur -> edges -> neg -> output
blur:  gaussian5   { sigma: 1.0 }
edges: edge_detect { scale: 0.5 }
neg:   invert      { enabled: true, strength: 1.0 }
"""


@pytest.fixture
def stage_dir(tmp_path):
    for f in ("edge_detect.stage.hip", "invert.stage.hip"):
        shutil.copy(os.path.join(SHADERS, f), tmp_path / f)
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    yield tmp_path
    rf.set_shader_path(old)


def test_unknown_type_without_a_shader_path_is_the_reference_error():
    old = rf.shader_path()
    rf.set_shader_path("")
    try:
        with pytest.raises(rf.RfError) as e:
            rf.Plan(rf.Config("input -> ee -> output\nee: edge_detect {}"))
        assert "no such filter" in str(e.value)
    finally:
        rf.set_shader_path(old)


def test_user_types_plan_and_fuse_with_built_in_nodes(stage_dir):
    p = rf.Plan(rf.Config(CHAIN))
    assert p.launches() == ["blur+edges+neg"]                 # one launch: user stages are row stages like any other
    info = p.launch_info()[0]
    assert info["radius"] == 2 + 1 + 0                        # gaussian5 + the 3x3 neighbourhood + the point op
    assert p.needs_jit() == [True]
    # the reference's one-launch-per-node schedule still works: every user node is compiled on its own
    q = rf.Plan(rf.Config(CHAIN), rf.RF_GRAPH_NO_FUSION)
    assert q.launches() == ["blur", "edges", "neg"] and q.needs_jit() == [False, True, True]
    # the diamond the reference's planner comments describe (pipeline_graph.rs:462-468), with the real edge_detect
    d = rf.Plan(rf.Config("input -> gaussian -> combination:input_image0\ninput -> edge_detect -> combination:input_image1\n"
                          "combination -> output\ngaussian: gaussian5 { sigma: 1.0 }\nedge_detect: edge_detect { scale: 1.0 }\ncombination: combination { mix: 0.5 }"))
    assert d.layers() == [["edge_detect", "gaussian"], ["combination"]] or len(d.launches()) == 1


@pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")
def test_user_stages_compile_for_gfx950_without_a_device(stage_dir):
    for text in (CHAIN, "input -> ee -> output\nee: edge_detect { scale: 2.0 }", "input -> nn -> output\nnn: invert { enabled: true, strength: 0.5 }"):
        p = rf.Plan(rf.Config(text))
        for fmt in (rf.RF_FORMAT_RGBA32F, rf.RF_FORMAT_RGBA8):
            assert p.jit_compile(fmt) > 4096


def test_parameters_are_the_members_of_struct_params(stage_dir):
    types = rf.registry_types()
    assert "edge_detect" not in types                         # the built-in registry is untouched
    # a parameter the struct does not declare is ignored with a warning status by rf_graph_set_param (render.rs:200-203);
    # here: the plan accepts the config, unknown members are simply not part of the type
    rf.Plan(rf.Config("input -> ee -> output\nee: edge_detect { scale: 1.0, nonsense: 3 }"))


def test_bad_stage_files_are_refused_with_a_reason(stage_dir):
    cases = {
        "noradius": "struct Params { float a; };\nRF_STAGE f4 apply(const Params& p, f4 c) { return c; }",
        "badtype": "struct Params { double a; };\nstatic constexpr int RADIUS = 0;\nRF_STAGE f4 apply(const Params& p, f4 c) { return c; }",
        "radius16": "struct Params { };\nstatic constexpr int RADIUS = 16;\nRF_STAGE f4 apply(const Params& p, f4 c) { return c; }",
        "noapply": "struct Params { };\nstatic constexpr int RADIUS = 0;\n",
    }
    for name, text in cases.items():
        (stage_dir / (name + ".stage.hip")).write_text(text)
        with pytest.raises(rf.RfError) as e:
            rf.Plan(rf.Config("input -> nn -> output\nnn: %s {}" % name))
        assert name + ".stage.hip" in str(e.value), str(e.value)


@pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")
def test_a_stage_that_does_not_compile_reports_the_compiler_message(stage_dir):
    (stage_dir / "broken.stage.hip").write_text("struct Params { float amt; };\nstatic constexpr int RADIUS = 0;\nRF_STAGE f4 apply(const Params& p, f4 c) { return c + undeclared_thing; }")
    p = rf.Plan(rf.Config("input -> nn -> output\nnn: broken { amt: 1.0 }"))        # parses: the text is only compiled at graph creation
    with pytest.raises(rf.RfError) as e:
        p.jit_compile()
    assert "undeclared_thing" in str(e.value) and "broken.stage.hip" in str(e.value)


def test_an_edited_file_is_a_new_stage(stage_dir):
    src = (stage_dir / "invert.stage.hip").read_text()
    a = rf.Plan(rf.Config("input -> nn -> output\nnn: invert { enabled: true, strength: 1.0 }"))
    m0 = rf.lib().rf_user_stage_mtime(b"invert")
    assert m0 > 0
    (stage_dir / "invert.stage.hip").write_text(src.replace("struct Params { bool enabled; float strength; };", "struct Params { bool enabled; float strength; float bias; };"))
    os.utime(stage_dir / "invert.stage.hip", ns=(m0 + 10 ** 9, m0 + 10 ** 9))
    b = rf.Plan(rf.Config("input -> nn -> output\nnn: invert { enabled: true, strength: 1.0, bias: 0.25 }"))
    assert rf.lib().rf_user_stage_mtime(b"invert") == m0 + 10 ** 9
    assert a.launches() == b.launches() == ["nn"]
