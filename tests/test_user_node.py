"""CPU: user NODES -- stage files that declare their images (RF_INPUTS / RF_OUTPUTS, rf_user.h): the counterpart of a .comp
file with several `uniform image2D` variables, each bound by its NAME (src/vulkan/shader.rs:151-153, vkutils.rs:159-183), one
allocated image per output binding (src/vulkan/pipeline_graph.rs:205-224), a name used on both sides written in place
(pipeline_graph.rs:228,:402-406).  Parsing, planning, error texts and the gfx950 code object (no device needed);
tests/test_gpu_user_node.py runs them."""
import glob
import os
import shutil
import sys

import pytest

import reforge_amd as rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

UNSHARP = """
input -> blur -> um:blurred_image
input -> um:input_image
um -> output
blur: gaussian9 { sigma: 2.0 }
um: unsharp_mask { amount: 1.5, threshold: 0.02 }
"""

# both outputs of the node are read: the mask is graded and mixed back over the sharpened image
UNSHARP_BOTH = """
input -> blur -> um:blurred_image
input -> um:input_image
um -> mm:input_image0
um:mask_image -> gg -> mm:input_image1
mm -> output
blur: gaussian5 { sigma: 1.0 }
um: unsharp_mask { amount: 0.8, threshold: 0.05 }
gg: colour_grade { slope: 1.5, offset: 0.0, saturation: 1.0 }
mm: combination { mix: 0.25 }
"""

# `image` is listed on both sides: ONE binding, the node writes its first input in place
TINT = """struct Params { float strength; };
static constexpr int RADIUS = 0;
RF_INPUTS(image, tint_image);
RF_OUTPUTS(image);
RF_STAGE void apply(const Params& p, const f4 (&in)[2], f4 (&out)[1])
{
    out[0] = make_float4(fmaf(p.strength, in[1].x - in[0].x, in[0].x), fmaf(p.strength, in[1].y - in[0].y, in[0].y),
                         fmaf(p.strength, in[1].z - in[0].z, in[0].z), in[0].w);
}
"""
TINT_GRAPH = """
input -> aa -> tt:image -> bb -> output
input -> cc -> tt:tint_image
aa: gaussian5 { sigma: 1.0 }
bb: sharpen { amount: 0.5 }
cc: colour_grade { slope: 0.5, offset: 0.1, saturation: 0.0 }
tt: tint { strength: 0.3 }
"""


# storage buffers by block type name (shader.rs:144-147): tone_curve fills ToneCurve, apply_curve reads it
CURVE = """
input -> tc -> ac -> output
tc:ToneCurve -> ac:ToneCurve
tc: tone_curve  { gamma: 0.6, lift: 0.05 }
ac: apply_curve { strength: 0.8 }
"""

# a user node that generates the K x K weights a built-in conv2d reads through its ConvWeights block
BOX_WEIGHTS = """struct Params { int ksize; float weight; };
static constexpr int RADIUS = 0;
RF_INPUTS(image);
RF_OUTPUTS(image);
RF_BUFFER_OUT(ConvWeights, 961);
RF_STAGE float fill(const Params& p, int i) { return i < p.ksize * p.ksize ? p.weight : 0.0f; }
RF_STAGE void apply(const Params& p, const f4 (&in)[1], f4 (&out)[1]) { out[0] = in[0]; }
"""
BOX_GRAPH = """
input -> bw:image -> cv -> output
bw:ConvWeights -> cv:ConvWeights
bw: box_weights { ksize: 5, weight: 0.04 }
cv: conv2d { ksize: 5 }
"""


@pytest.fixture
def stage_dir(tmp_path):
    for f in ("unsharp_mask.stage.hip", "tone_curve.stage.hip", "apply_curve.stage.hip", "local_contrast.stage.hip"):
        shutil.copy(os.path.join(SHADERS, f), tmp_path / f)
    (tmp_path / "box_weights.stage.hip").write_text(BOX_WEIGHTS)
    (tmp_path / "tint.stage.hip").write_text(TINT)
    old = rf.shader_path()
    rf.set_shader_path(str(tmp_path))
    yield tmp_path
    rf.set_shader_path(old)


def test_images_are_bound_by_the_names_the_file_declares(stage_dir):
    L = rf.lib()
    assert [L.rf_registry_binding(b"unsharp_mask", n) for n in (b"input_image", b"blurred_image", b"output_image", b"mask_image", b"nonsense")] == [0, 1, 2, 3, -1]
    assert [L.rf_registry_binding(b"tint", n) for n in (b"image", b"tint_image")] == [0, 1]
    p = rf.Plan(rf.Config(UNSHARP))
    um = p.launch_info()[1]
    assert um["label"] == "um" and um["radius"] == 0
    assert um["inputs"] == ["rf:file-input", "blur:output_image"]          # declaration order, whatever order the config wires them in
    assert um["outputs"] == ["rf:final-output"]                            # mask_image is not wired: not allocated, not stored
    assert p.needs_jit() == [False, True]
    flipped = rf.Plan(rf.Config(UNSHARP.replace("input -> blur -> um:blurred_image\ninput -> um:input_image", "input -> um:input_image\ninput -> blur -> um:blurred_image")))
    assert flipped.launch_info()[1]["inputs"] == um["inputs"]


def test_every_output_binding_gets_an_image_and_the_node_is_never_fused(stage_dir):
    p = rf.Plan(rf.Config(UNSHARP_BOTH))
    info = {l["label"]: l for l in p.launch_info()}
    assert set(info) == {"blur", "um", "gg", "mm"}                          # a user node keeps a launch of its own; so do its neighbours here
    assert len(info["um"]["outputs"]) == 2 and len(set(info["um"]["outputs"])) == 2
    assert info["mm"]["inputs"][0] == info["um"]["outputs"][0] and info["gg"]["inputs"] == [info["um"]["outputs"][1]]
    # the over-fetch schedule of a row-strip partition treats it as the point op it is
    need_src, need_dst, need_input, _ghost = p.halo_schedule(exchange=False)
    k = [l["label"] for l in p.launch_info()].index("um")
    assert need_src[k] == need_dst[k] == 0 and need_input == 2              # only the gaussian5 in front of it reads ghost rows


def test_a_name_on_both_sides_is_written_in_place(stage_dir):
    p = rf.Plan(rf.Config(TINT_GRAPH))
    info = {l["label"]: l for l in p.launch_info()}
    tt = info["tt"]
    assert tt["outputs"] == [tt["inputs"][0]] == info["aa"]["outputs"]     # pipeline_graph.rs:402-406: the output IS the input image
    assert tt["inputs"][1] == info["cc"]["outputs"][0] and tt["inputs"][1] != tt["inputs"][0]


def test_plans_of_user_node_graphs_match_the_restatement(stage_dir):
    """the planner's layers / aliases / allocated images / buffers for graphs that hold user nodes against the restatement of
    order_by_execution and reusable_image_remapping (oracle/graph.py), which is handed the tables "reflection" of the files yields"""
    from oracle import graph as og
    L = rf.lib()
    tables = {
        "unsharp_mask": {"images": {n: L.rf_registry_binding(b"unsharp_mask", n.encode()) for n in ("input_image", "blurred_image", "output_image", "mask_image")},
                         "params": {"amount": "f32", "threshold": "f32"}},
        "tint": {"images": {n: L.rf_registry_binding(b"tint", n.encode()) for n in ("image", "tint_image")}, "params": {"strength": "f32"}},
        "tone_curve": {"images": {"input_image": 0, "output_image": 1}, "params": {"gamma": "f32", "lift": "f32"},
                       "buffers": {"ToneCurve": (L.rf_registry_buffer_binding(b"tone_curve", b"ToneCurve"), 256 * 4)}},
        "apply_curve": {"images": {"input_image": 0, "output_image": 1}, "params": {"strength": "f32"},
                        "buffers": {"ToneCurve": (L.rf_registry_buffer_binding(b"apply_curve", b"ToneCurve"), 256 * 4)}},
    }
    added = [k for k in tables if k not in og.NODE_TYPES]
    og.NODE_TYPES.update(tables)
    try:
        for text in (UNSHARP, UNSHARP_BOTH, TINT_GRAPH, CURVE,
                     "input -> um:input_image\ninput -> aa -> bb -> um:blurred_image\num:mask_image -> cc -> output\num: unsharp_mask {}\naa: gaussian5 {}\nbb: sharpen {}\ncc: sharpen {}"):
            p = rf.Plan(rf.Config(text), rf.RF_GRAPH_NO_FUSION)
            infos = og.synthesize(og.parse_config(text))
            layers = og.order_by_execution(infos)
            assert p.layers() == layers, text
            assert p.aliases() == og.reusable_image_remapping(layers, infos), text
            assert p.launches() == [n for l in layers for n in l]
    finally:
        for k in added:
            del og.NODE_TYPES[k]


def test_wiring_errors_name_the_image_variable(stage_dir):
    cases = {
        "input -> um:input_image\num -> output\num: unsharp_mask {}": "needs an image wired to blurred_image",
        "input -> um:input_image\ninput -> um:blurred_image\ninput -> um:mask_image\num -> output\num: unsharp_mask {}": "mask_image is an output image of unsharp_mask",
        "input -> um:input_image\ninput -> um:blurred_image\num:blurred_image -> output\num: unsharp_mask {}": "blurred_image is an input image of unsharp_mask",
        "input -> um:input_image\ninput -> um:nonsense\num -> output\num: unsharp_mask {}": "no binding named: nonsense",
    }
    for text, want in cases.items():
        with pytest.raises(rf.RfError) as e:
            rf.Plan(rf.Config(text)).halo_schedule()          # what the kernels cannot execute is reported by everything that needs the launch list
        assert want in str(e.value), (text, str(e.value))


# a node that reads TWO inputs through windows (RADIUS 1 with declared images: not a row stage): a gradient of the guide added to the base
GUIDED = """struct Params { float strength; };
static constexpr int RADIUS = 1;
RF_INPUTS(base_image, guide_image);
RF_OUTPUTS(output_image);
RF_STAGE void apply(const Params& p, const Window (&in)[2], f4 (&out)[1])
{
    const f4 c = in[0].at(0, 0), e = in[1].at(1, 0), w = in[1].at(-1, 0), s = in[1].at(0, 1), n = in[1].at(0, -1);
    out[0] = make_float4(fmaf(p.strength, (e.x - w.x) + (s.x - n.x), c.x), fmaf(p.strength, (e.y - w.y) + (s.y - n.y), c.y),
                         fmaf(p.strength, (e.z - w.z) + (s.z - n.z), c.z), c.w);
}
"""
WINDOW_GRAPH = """
input -> gg -> lc -> gd:base_image
input -> sh -> gd:guide_image
gd -> output
gg: gaussian5 { sigma: 1.0 }
lc: local_contrast { amount: 0.8 }
sh: sharpen { amount: 0.4 }
gd: guided { strength: 0.25 }
"""


def test_a_stage_of_radius_two_or_more_is_a_node_that_reads_through_windows(stage_dir):
    (stage_dir / "guided.stage.hip").write_text(GUIDED)
    p = rf.Plan(rf.Config(WINDOW_GRAPH))
    info = {l["label"]: l for l in p.launch_info()}
    assert set(info) == {"gg", "lc", "sh", "gd"}                            # window nodes keep launches of their own
    assert info["lc"]["radius"] == 2 and info["gd"]["radius"] == 1 and len(info["gd"]["inputs"]) == 2
    assert info["lc"]["inputs"] != info["lc"]["outputs"]                    # never planned in place: it reads its neighbours
    # row strips: the ghost rows a window node reads are scheduled like any stencil's (over-fetch: cumulative; exchange: per launch)
    labels = [l["label"] for l in p.launch_info()]
    need_src, need_dst, need_input, ghost = p.halo_schedule(exchange=False)
    assert need_input == 2 + 2 + 1 and ghost == 5 and need_src[labels.index("lc")] == 3 and need_dst[labels.index("lc")] == 1
    xs, _xd, _ni, xg = p.halo_schedule(exchange=True)
    assert xs[labels.index("lc")] == 2 and xs[labels.index("gd")] == 1 and xg == 2
    if rf.lib().rf_jit_available():
        assert p.jit_compile(rf.RF_FORMAT_RGBA32F) > 4096 and p.jit_compile(rf.RF_FORMAT_RGBA8) > 4096


def test_storage_buffers_are_found_by_their_block_type_name(stage_dir):
    L = rf.lib()
    assert L.rf_registry_buffer_binding(b"tone_curve", b"ToneCurve") == 2 and L.rf_registry_buffer_binding(b"apply_curve", b"ToneCurve") == 2
    assert L.rf_registry_buffer_binding(b"apply_curve", b"Nonsense") == -1
    p = rf.Plan(rf.Config(CURVE))
    assert p.launches() == ["tc", "ac"] and p.layers() == [["tc"], ["ac"]]      # the buffer edge orders the nodes like an image edge
    assert p.buffers() == {"tc:ToneCurve": 256 * 4}
    assert p.needs_jit() == [True, True]
    # the buffer edge alone orders them: no image edge between the writer and the reader
    q = rf.Plan(rf.Config("input -> tc\ninput -> ac -> output\ntc:ToneCurve -> ac:ToneCurve\ntc -> mm:input_image0\nac -> mm:input_image1\nmm -> output\n"
                          "tc: tone_curve {}\nac: apply_curve {}\nmm: combination { mix: 0.5 }".replace("input -> ac -> output", "input -> ac")))
    assert q.layers() == [["tc"], ["ac"], ["mm"]]
    # a built-in reader of a user-written block: conv2d takes its K x K weights from what box_weights fills
    b = rf.Plan(rf.Config(BOX_GRAPH))
    assert b.launches() == ["bw", "cv"] and b.buffers() == {"bw:ConvWeights": 961 * 4}
    cases = {
        "input -> ac -> output\nac: apply_curve { strength: 1.0 }": "needs a storage buffer wired to ToneCurve",
        "input -> tc -> ac -> output\ntc:ToneCurve -> ac:ToneCurve\nac:ToneCurve -> tc:ToneCurve\ntc: tone_curve {}\nac: apply_curve {}": "",
    }
    for text, want in cases.items():
        with pytest.raises(rf.RfError) as e:
            rf.Plan(rf.Config(text)).halo_schedule()
        assert want in str(e.value), str(e.value)


def test_bad_declarations_are_refused_with_a_reason(stage_dir):
    body = "RF_STAGE void apply(const Params& p, const f4 (&in)[1], f4 (&out)[1]) { out[0] = in[0]; }"
    cases = {
        "inplacewin": ("struct Params { };\nstatic constexpr int RADIUS = 1;\nRF_INPUTS(image);\nRF_OUTPUTS(image);\n" + body, "cannot be written in place"),
        "radius16": ("struct Params { };\nstatic constexpr int RADIUS = 16;\nRF_INPUTS(aa_image);\n" + body, "RADIUS must be"),
        "toomany": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_INPUTS(a1, a2, a3, a4, a5);\n" + body, "1 to 4 image names"),
        "twice": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_OUTPUTS(o1, o1);\n" + body, "lists `o1` twice"),
        "notaname": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_INPUTS(a b);\n" + body, "is not an image variable name"),
        "empty": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_INPUTS();\n" + body, "1 to 4 image names"),
        "nocount": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_BUFFER_IN(Curve);\n" + body, "number of floats"),
        "toobig": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_BUFFER_IN(Curve, 70000);\n" + body, "number of floats"),
        "nofill": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_BUFFER_OUT(Curve, 16);\n" + body, "RF_BUFFER_OUT needs"),
        "both": ("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_BUFFER_IN(Curve, 16);\nRF_BUFFER_OUT(Curve, 16);\nRF_STAGE float fill(const Params& p, int i) { return 0.f; }\n" + body,
                 "both read and written"),
    }
    for name, (text, want) in cases.items():
        (stage_dir / (name + ".stage.hip")).write_text(text)
        with pytest.raises(rf.RfError) as e:
            rf.Plan(rf.Config("input -> nn -> output\nnn: %s {}" % name))
        assert name + ".stage.hip" in str(e.value) and want in str(e.value), str(e.value)


@pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")
def test_user_nodes_compile_for_gfx950_and_move_every_texel_once(stage_dir, tmp_path, monkeypatch):
    """the code object of user_node_kernel<Px, Stage> (rf_user_dev.h): per texel ONE load per input image and ONE store per
    output image, no scratch, no LDS, no barrier -- the launch is bound by HBM, (NI + NO) x W x H x bytes-per-pixel"""
    import isa_obj
    cache = tmp_path / "cache"
    monkeypatch.setenv("RF_JIT_CACHE_DIR", str(cache))
    for text, ni, no in ((UNSHARP_BOTH, 2, 2), (TINT_GRAPH, 2, 1), (CURVE, 1, 1)):
        for fmt, ld, st in ((rf.RF_FORMAT_RGBA32F, "global_load_dwordx4", "global_store_dwordx4"), (rf.RF_FORMAT_RGBA8, "global_load_dword", "global_store_dword")):
            before = set(glob.glob(str(cache / "*.hsaco")))
            assert rf.Plan(rf.Config(text)).jit_compile(fmt) > 2048
            new = sorted(set(glob.glob(str(cache / "*.hsaco"))) - before)
            nodes = [f for f in new if open(f[:-6] + ".name").read().startswith("_ZN2rf16user_node_kernel")]
            fills = [f for f in new if open(f[:-6] + ".name").read().startswith("_ZN2rf16user_fill_kernel")]
            if text is CURVE:
                # two user nodes; the curve lookups of apply_curve are extra loads from the 1 KiB buffer (L1/L2-resident), not image traffic
                assert len(nodes) == 2 and len(fills) <= 1, new
                for f in fills:
                    (fname, fins), = [(n, i) for n, i in isa_obj.functions(f).items() if n.startswith("_ZN2rf16user_fill_kernel")]
                    fops = [i.op for i in fins]
                    assert fops.count("global_store_dword") == 1 and not any(o.startswith(("scratch_", "ds_", "global_load")) for o in fops), fname
                continue
            assert len(nodes) == 1, new
            (name, ins), = [(n, i) for n, i in isa_obj.functions(nodes[0]).items() if n.startswith("_ZN2rf16user_node_kernel")]
            ops = [i.op for i in ins]
            assert not any(o.startswith(("scratch_", "ds_", "buffer_", "flat_")) for o in ops) and "s_barrier" not in ops, name
            # loads: one per input image, or narrower pieces where apply() ignores a channel (the compiler drops dead channels)
            loads = [o for o in ops if o.startswith("global_load")]
            dwords = sum({"global_load_dword": 1, "global_load_dwordx2": 2, "global_load_dwordx3": 3, "global_load_dwordx4": 4}[o] for o in loads)
            assert ni <= len(loads) <= 2 * ni and dwords <= ni * (4 if ld.endswith("x4") else 1), (name, loads)
            assert [o for o in ops if o.startswith("global_store")] == [st] * no, (name, [o for o in ops if o.startswith("global_")])


@pytest.mark.skipif(not rf.lib().rf_jit_available(), reason="libhiprtc cannot be loaded")
def test_a_node_whose_apply_does_not_match_its_declaration_reports_the_compiler_message(stage_dir):
    (stage_dir / "wrongarity.stage.hip").write_text("struct Params { };\nstatic constexpr int RADIUS = 0;\nRF_INPUTS(aa_image, bb_image);\n"
                                                    "RF_STAGE void apply(const Params& p, const f4 (&in)[1], f4 (&out)[1]) { out[0] = in[0]; }")
    p = rf.Plan(rf.Config("input -> nn:aa_image\ninput -> gg -> nn:bb_image\nnn -> output\nnn: wrongarity {}\ngg: gaussian5 { sigma: 1.0 }"))
    with pytest.raises(rf.RfError) as e:
        p.jit_compile()
    assert "wrongarity.stage.hip" in str(e.value) or "apply" in str(e.value)


WINDOW_ISA_CHILD = r"""
import sys
if sys.argv[2] == "torch":
    import torch  # noqa: F401  (first: the process then compiles with the libhiprtc PyTorch bundles, an older compiler build)
import collections, glob, os
os.environ["RF_JIT_CACHE_DIR"] = sys.argv[1]
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, os.path.join(sys.argv[3], "scripts"))
import reforge_amd as rf
import isa_obj
rf.set_shader_path(os.path.join(sys.argv[3], "shaders"))
for text in ("input -> lc -> output\nlc: local_contrast { amount: 0.8 }", "input -> st -> output\nst: streak { amount: 0.6 }"):
    p = rf.Plan(rf.Config(text))
    p.jit_compile(rf.RF_FORMAT_RGBA32F)
    p.jit_compile(rf.RF_FORMAT_RGBA8)
for f in sorted(glob.glob(sys.argv[1] + "/*.hsaco")):
    for n, ins in isa_obj.functions(f).items():
        if "user_node_kernel" not in n:
            continue
        c = collections.Counter(i.op for i in ins)
        print("KERNEL", n, sum(v for k, v in c.items() if k.startswith("ds_read")), sum(v for k, v in c.items() if k.startswith("scratch_")),
              sum(v for k, v in c.items() if k.startswith(("global_load", "flat_"))), c.get("s_barrier", 0), flush=True)
os._exit(0)
"""


@pytest.mark.parametrize("first", ["plain", "torch"])
def test_window_node_kernels_stage_their_tiles_in_lds_under_both_compilers(tmp_path, first):
    """rf_user_dev.h: a node that reads through windows stages its tile in LDS behind ONE barrier, loads every input texel from
    memory once per tile (one load instruction in the fill loop), keeps nothing in scratch; RADIUS 2 slides a register copy of
    the window -- 5 LDS reads per output, not 25 -- which must hold under the compiler build PyTorch bundles too (bench.py's
    process imports torch first; that build left the copy in scratch until it was indexed with constants only: 1.3 ms instead
    of 47 us at 4K, and no parity test could notice)."""
    import subprocess
    if first == "torch":
        import importlib.util
        if importlib.util.find_spec("torch") is None:
            pytest.skip("no PyTorch here")
    child = tmp_path / "child.py"
    child.write_text(WINDOW_ISA_CHILD)
    cache = tmp_path / "cache"
    cache.mkdir()
    r = subprocess.run([sys.executable, str(child), str(cache), first, ROOT], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [l.split() for l in r.stdout.splitlines() if l.startswith("KERNEL")]
    assert len(rows) == 4, r.stdout
    for _k, name, ds_reads, scratch, loads, barriers in rows:
        assert int(scratch) == 0 and int(barriers) == 1 and int(loads) == 1, (name, ds_reads, scratch, loads, barriers)
        assert int(ds_reads) <= 80, (name, ds_reads)      # local_contrast: 8 outputs x 5 + the first window; streak: 29 taps x 4 outputs, some paired
