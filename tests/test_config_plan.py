"""CPU: the product's config parser and planner (through the C ABI, host-only entry
points) against the reference's documented semantics and
against the Python restatement in oracle/graph.py."""
import pytest

import reforge_amd as rf
from oracle import graph as og
from oracle import pixel
from tests import util

NF = rf.RF_GRAPH_NO_FUSION


def both(text, expects_input=True):
    return rf.Config(text, expects_input).nodes(), og.parse_config(text, expects_input)


def same_config(text, expects_input=True):
    prod, ora = both(text, expects_input)
    assert set(prod) == set(ora.graph_pipelines)
    for name, node in prod.items():
        assert node["inputs"] == ora.graph_pipelines[name]["inputs"], name
        assert node["outputs"] == ora.graph_pipelines[name]["outputs"], name
        assert node["type"] == ora.type_of(name)
        assert node["params"] == ora.params_of(name)
    return prod


def test_default_graph_kat():
    """"input -> passthrough -> output" (render.rs:115)."""
    n = same_config("input -> passthrough -> output")
    assert n == {"passthrough": {"type": "passthrough", "inputs": [("rf:file-input", "input_image")],
                                 "outputs": [("rf:final-output", "output_image")], "params": {}}}


def test_descriptor_suffix_semantics():
    """config.rs:169-186: a mid-chain `X:foo` uses foo as BOTH its input and its output
    descriptor; the next node reads resource `X:foo`."""
    n = same_config("input -> aa:image -> bb -> output")
    assert n["aa"]["inputs"] == [("rf:file-input", "image")]
    assert n["aa"]["outputs"] == [("aa:image", "image")]
    assert n["bb"]["inputs"] == [("aa:image", "input_image")]
    assert n["bb"]["outputs"] == [("rf:final-output", "output_image")]


def test_instances_and_values():
    text = """
    // a comment between expressions
    input -> blur -> output
    blur: gaussian { sigma: 2.5, radius: 3, flag: true, neg: -0.25, sigma: 1.5 }
    /* block */
    other: sharpen {}
    """
    n = same_config(text)
    assert n["blur"]["type"] == "gaussian"
    assert n["blur"]["params"] == {"sigma": "1.5", "radius": "3", "flag": "true", "neg": "-0.25"}   # last insert wins


def test_multiple_graph_expressions_accumulate():
    n = same_config(util.DIAMOND)
    assert n["mixer"]["inputs"] == [("blur:output_image", "input_image0"), ("sharp:output_image", "input_image1")]
    assert n["mixer"]["outputs"] == [("rf:final-output", "output_image")]


@pytest.mark.parametrize("text,expects_input", [
    ("", True), ("   \n\t", True),                    # config.rs:99-102
    ("input -> aa", True),                            # 'output' never used, :202
    ("input -> aa -> output", False),                 # 'input' without an input image, :201
    ("input -> output", True),                        # empty graph, :200
    ("input->aa->output", True),                      # '-' belongs to the identifier regex: "input-" then '>'
    ("a -> output", True),                            # identifiers need two characters (:81)
    ("input -> aa -> output\naa: passthrough { }", True),     # "{ }" is not the "{}" token
    ("input -> aa -> output\naa: gaussian { sigma: -3 }", True),   # no negative integers (:75-76)
    ("input -> aa -> output\naa: gaussian { sigma: 1e3 }", True),  # no exponents
    ("input -> aa // trailing\n -> output", True),    # a comment ends the graph expression
    ("aa", True), ("aa:bb", True), ("input -> -> output", True),
    ("input -> aa -> output\naa: gaussian { sigma 1.0 }", True),
])
def test_rejected_configs(text, expects_input):
    with pytest.raises(rf.RfError) as e:
        rf.Config(text, expects_input)
    assert e.value.status == 2          # RF_ERR_CONFIG
    with pytest.raises(og.ConfigError):
        og.parse_config(text, expects_input)


def test_block_comment_longest_match_quirk():
    """config_grammar.lalrpop:27: the regex accepts any text between the first "/*" and
    the LAST "*/", and the lexer takes the longest match -- everything between two
    block comments is swallowed.  Kept as the reference behaves."""
    text = "input -> aa -> output\n/* one */ aa: gaussian { sigma: 9.0 } /* two */"
    n = same_config(text)
    assert n["aa"]["type"] == "aa" and n["aa"]["params"] == {}


def test_single_shader_mode():
    """config.rs:77-90"""
    n = rf.Config(single="sharpen").nodes()
    assert list(n) == ["sharpen"] and n["sharpen"]["inputs"] == [("rf:file-input", "input_image")]
    n = rf.Config(single="sharpen", expects_input=False).nodes()
    assert n["sharpen"]["inputs"] == [] and n["sharpen"]["outputs"] == [("rf:final-output", "output_image")]


# ---- planner -----------------------------------------------------------------------
def plans(text):
    p = rf.Plan(rf.Config(text), NF)
    cfg = og.parse_config(text)
    infos = og.synthesize(cfg)
    layers = og.order_by_execution(infos)
    reuse = og.reusable_image_remapping(layers, infos)
    return p, layers, reuse


PLAN_CASES = [
    "input -> passthrough -> output",
    util.CHAIN3, util.CHAIN5, util.DIAMOND,
    "input -> aa -> bb -> cc -> dd -> output\naa: passthrough {}\nbb: sharpen {}\ncc: passthrough {}\ndd: sharpen {}",
    "input -> gaussian5 -> colour_grade:image -> sharpen -> output",
    "input -> aa:image -> bb:image -> output\naa: grade {}\nbb: grade {}",
    "input -> aa -> output\ninput -> bb -> output\naa: sharpen {}\nbb: gaussian5 {}",
    util.SPLIT2,
    # only ONE of the two outputs wired; and a split whose outputs feed a chain each before the join
    "input -> sp\nsp:chroma_image -> sharpen -> output\nsp: split_luma {}",
    "input -> sp\nsp:luma_image -> aa -> bb -> mx:input_image0\nsp:chroma_image -> mx:input_image1\nmx -> output\nsp: split_luma {}\naa: sharpen {}\nbb: gaussian5 {}\nmx: combination {}",
]


@pytest.mark.parametrize("text", PLAN_CASES)
def test_plan_matches_restatement(text):
    p, layers, reuse = plans(text)
    assert p.layers() == layers
    assert p.aliases() == reuse
    g = og.GraphOracle(text, 4, 4, util.F32)
    assert p.images() == g.allocated_images()
    assert p.launches() == [n for l in layers for n in l]
    for r in list(reuse) + ["rf:final-output", "rf:file-input"]:
        assert p.resolve(r) == og._remap(r, reuse)


def test_a_node_with_two_output_images_kat():
    """pipeline_graph.rs:205-224: one image per output binding of every node.  split_luma writes luma_image and chroma_image."""
    n = same_config(util.SPLIT2)
    assert n["sp"]["inputs"] == [("rf:file-input", "input_image")]
    assert n["sp"]["outputs"] == [("sp:luma_image", "luma_image"), ("sp:chroma_image", "chroma_image")]
    p = rf.Plan(rf.Config(util.SPLIT2), NF)
    assert p.layers() == [["sp"], ["cg", "lg"], ["mx"]]
    info = {l["label"]: l for l in p.launch_info()}
    assert len(info["sp"]["outputs"]) == 2 and len(set(info["sp"]["outputs"])) == 2 and info["sp"]["output"] == info["sp"]["outputs"][0]
    assert info["sp"]["outputs"] == [p.resolve("sp:luma_image"), p.resolve("sp:chroma_image")]       # binding order
    assert info["lg"]["inputs"] == [p.resolve("sp:luma_image")] and info["cg"]["inputs"] == [p.resolve("sp:chroma_image")]
    # with fusion: the split keeps a launch of its own, its consumers are planned as before
    q = rf.Plan(rf.Config(util.SPLIT2))
    assert q.launches()[0] == "sp" and len(q.launch_info()[0]["outputs"]) == 2
    # the strip schedule covers BOTH outputs: the gaussian behind luma_image needs 2 ghost rows of it
    ns, nd, need_input, ghost = p.halo_schedule(exchange=False)
    assert nd[0] == 2 and need_input == 2 and ghost == 2
    with pytest.raises(rf.RfError):
        rf.Plan(rf.Config("input -> sp:nonsense -> output\nsp: split_luma {}"))


def test_diamond_layers_kat():
    """pipeline_graph.rs:462-468: two independent branches form one layer, the join the next."""
    p, _, _ = plans(util.DIAMOND)
    assert p.layers() == [["blur", "sharp"], ["mixer"]]


def test_four_chain_alias_kat():
    """A->B->C->D: C reuses A's image, the final output reuses B's."""
    p, _, _ = plans(PLAN_CASES[4])
    assert p.aliases() == {"cc:output_image": "aa:output_image", "rf:final-output": "bb:output_image"}
    assert p.images() == ["aa:output_image", "bb:output_image", "rf:file-input"]     # input + 2 ping-pong images


def test_point_op_alias_kat():
    """pipeline_graph.rs:400-411: same binding for input and output => in place."""
    p, _, _ = plans(PLAN_CASES[5])
    assert p.aliases() == {"colour_grade:image": "gaussian5:output_image"}
    assert p.resolve("colour_grade:image") == "gaussian5:output_image"


def test_file_input_is_never_recycled():
    p, _, _ = plans(util.CHAIN5)
    assert "rf:file-input" in p.images() and "rf:file-input" not in p.aliases().values()


def test_cycle_is_rejected():
    """pipeline_graph.rs:487-490"""
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> aa -> bb -> aa -> output\naa: sharpen {}\nbb: sharpen {}"), NF)
    assert e.value.status == 3 and "Graph incorrectly constructed" in str(e.value)


def test_unknown_type_and_binding():
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> nosuchfilter -> output"), NF)
    assert e.value.status == 3
    with pytest.raises(rf.RfError) as e:          # vkutils.rs:179
        rf.Plan(rf.Config("input -> sharpen:nosuchimage -> output"), NF)
    assert "has no binding named: nosuchimage" in str(e.value)


def test_fusion_groups():
    p = rf.Plan(rf.Config(util.CHAIN3), 0)
    assert p.launches() == ["blur+grade+sharp"] and p.layers() == [["blur+grade+sharp"]]
    assert p.images() == ["rf:file-input", "rf:final-output"] and p.aliases() == {}
    p = rf.Plan(rf.Config(util.CHAIN5), 0)
    assert p.launches() == ["blur+grade+sharp+wide+finish"]        # the whole BASELINE 5-stage chain is one launch
    assert p.halo_schedule(True) == ([7], [0], 0, 7)               # one exchange of 2+1+4 rows instead of three
    assert p.halo_schedule(False) == ([7], [0], 7, 7)              # over-fetch: the strip's input carries the 7 rows
    # any other chain of fusable nodes is ONE launch too: its kernel is compiled when the graph is created (rf_jit.cpp) ...
    other = util.CHAIN5_SPLIT
    p = rf.Plan(rf.Config(other), 0)
    if rf.lib().rf_jit_available():
        assert p.launches() == ["blur+grade+sharp+wide+finish"] and p.needs_jit() == [True]
    # ... and with RF_GRAPH_NO_JIT it splits greedily into the longest prefixes the ahead-of-time catalogue holds
    p = rf.Plan(rf.Config(other), rf.RF_GRAPH_NO_JIT)
    assert p.launches() == ["blur+grade+sharp", "wide+finish"] and p.needs_jit() == [False, False]
    # passthrough and gaussian{radius} nodes join chains; the admission rule keeps what a lane's registers can hold
    mixed = "input -> aa -> bb -> cc -> dd -> output\naa: gaussian { sigma: 1.5, radius: 3 }\nbb: passthrough {}\ncc: sharpen { amount: 0.4 }\ndd: grade {}"
    if rf.lib().rf_jit_available():
        assert rf.Plan(rf.Config(mixed), 0).launches() == ["aa+bb+cc+dd"]
        wide = "input -> aa -> bb -> output\naa: gaussian { sigma: 4.0, radius: 12 }\nbb: gaussian { sigma: 4.0, radius: 12 }"
        assert rf.Plan(rf.Config(wide), 0).launches() == ["aa", "bb"]         # two radius-12 windows do not fit: not fused
    # a fused chain is planned as one node: the aliasing plan is recomputed on the fused
    # graph, so its output can never land on the image it reads
    assert p.resolve("rf:final-output") != "rf:file-input"
    # a fork/join whose branches descend from one image is ONE launch: both branches ride the stage chain as a pair of rows
    d = rf.Plan(rf.Config(util.DIAMOND), 0)
    assert d.launches() == ["blur+sharp+mixer"] and d.needs_jit() == [False]        # (this very diamond is in the catalogue)
    assert d.launch_info()[0]["inputs"] == ["rf:file-input"] and d.launch_info()[0]["radius"] == 2        # pre + max(2, 1) + post: the branches run side by side
    assert d.images() == ["rf:file-input", "rf:final-output"]                       # neither branch result is materialised
    assert rf.Plan(rf.Config(util.DIAMOND), NF).launches() == ["blur", "sharp", "mixer"]
    # a branch image that somebody else reads keeps the fork/join unfused
    assert "blur" in rf.Plan(rf.Config(util.DIAMOND.replace("mixer -> output", "mixer -> m2:input_image0\nblur -> m2:input_image1\nm2 -> output") + "\nm2: combination { mix: 0.5 }"), 0).launches()
    # an image that two nodes read is materialised
    fork = "input -> aa -> bb -> output\naa -> cc -> output\naa: gaussian5 {}\nbb: grade {}\ncc: grade {}"
    assert "aa" in rf.Plan(rf.Config(fork), 0).launches()


def test_registry():
    assert rf.registry_binding("passthrough", "input_image") == 0      # passthrough.comp:4
    assert rf.registry_binding("passthrough", "output_image") == 1     # passthrough.comp:5
    assert rf.registry_binding("passthrough", "image") == -1
    assert rf.registry_binding("colour_grade", "image") == 2
    assert set(og.NODE_TYPES) == set(rf.registry_types())


def test_get_dim():
    """utils.rs:56-74"""
    assert rf.get_dim(800, 600) == (800, 600)
    assert rf.get_dim(800, 600, 400, None) == (400, 300)
    assert rf.get_dim(800, 600, None, 300) == (400, 300)
    assert rf.get_dim(800, 600, 123, 45) == (123, 45)
    assert rf.get_dim(1920, 1080, 1000, None) == (1000, 562)


def test_strip_rows_partition():
    for H, world in ((2160, 8), (16384, 8), (17, 4), (5, 5), (1080, 3)):
        rows = [rf.strip_rows(H, world, r) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == H
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in rows]
        assert max(sizes) - min(sizes) <= 1


TWO_LEVEL_ALIAS = """
input -> n00 -> n01 -> mx:input_image0
n00 -> n02 -> n03 -> n04:image -> n05 -> mx:input_image1
mx -> output
n00: gaussian5 { sigma: 1.09 }
n01: sharpen { amount: 0.34 }
n02: passthrough {}
n03: sharpen { amount: 0.65 }
n04: colour_grade { slope: 0.85, offset: -0.055, saturation: 1.23 }
n05: conv2d { ksize: 5, sigma: 1.79 }
mx: combination { mix: 0.87 }
"""


def test_alias_chain_is_followed_when_freeing_images():
    """The one deliberate difference from pipeline_graph.rs:358-427 (found by the random-graph
    test): n03 recycles n00's image, n04 runs in place on it (n04:image -> n03:output_image ->
    n00:output_image).  The reference tests one alias level (:364-369), sees nobody using
    n00:output_image at n05's layer, frees it and hands it to n05 -- a 5x5 convolution then
    writes the image it reads.  The plan here follows the chain and gives n05 another image."""
    cfg = og.parse_config(TWO_LEVEL_ALIAS)
    infos = og.synthesize(cfg)
    layers = og.order_by_execution(infos)
    literal = og.reusable_image_remapping(layers, infos, literal=True)
    assert og._remap("n05:output_image", literal) == og._remap("n04:image", literal) == "n00:output_image"   # the hazard
    ours = og.reusable_image_remapping(layers, infos)
    assert og._remap("n04:image", ours) == "n00:output_image"
    assert og._remap("n05:output_image", ours) != "n00:output_image"
    p = rf.Plan(rf.Config(TWO_LEVEL_ALIAS), NF)
    assert p.aliases() == ours and p.layers() == layers
    assert len(p.launch_info()) == 7                     # every node launches: nothing "would run a stencil in place"
    # wherever no image is read through a two-level alias the two agree (all the hand-written cases)
    for text in PLAN_CASES:
        i2 = og.synthesize(og.parse_config(text))
        l2 = og.order_by_execution(i2)
        assert og.reusable_image_remapping(l2, i2) == og.reusable_image_remapping(l2, i2, literal=True)


def test_parser_differential_fuzz():
    """Token soup through both parsers (the C++ one behind the ABI and the Python restatement of
    config_grammar.lalrpop): they accept and reject the same texts and build the same graph --
    and the C++ one survives everything."""
    import numpy as np
    vocab = ["input", "output", "aa", "bb", "cc", "blur", "a", "x1", "gaussian5", "sharpen", "colour_grade", "->", "->", "->", ":",
             "{", "}", "{}", ",", "sigma", "amount", "1.5", "2", "-0.5", "-3", "true", "false", "1e3", "// note\n", "/* c */", "\n",
             "\n", " ", "  ", "\t", "image", "input_image", "_x", "a-b", "-", ">", "*/", "/*", "é", "0", ".5", "5."]
    rng = np.random.RandomState(77)
    idents = ["aa", "bb", "cc", "blur", "x1", "_x", "a-b", "gaussian5", "sharpen", "grade", "input_image", "image", "n0"]
    values = ["1.5", "2", "-0.5", "true", "false", "0", "10.25", "-3", "1e3", ".5"]

    def pick(xs):
        return xs[rng.randint(len(xs))]

    def well_formed():
        exprs = []
        for _ in range(rng.randint(1, 4)):
            kind = rng.randint(4)
            if kind == 0:
                mid = [pick(idents) + (":" + pick(idents) if rng.rand() < 0.25 else "") for _ in range(rng.randint(1, 4))]
                exprs.append(pick(["input", pick(idents)]) + " -> " + " -> ".join(mid) + pick([" -> output", " -> output", ""]))
            elif kind == 1:
                kv = ["%s: %s" % (pick(idents), pick(values)) for _ in range(rng.randint(0, 3))]
                exprs.append("%s: %s %s" % (pick(idents), pick(idents), "{ " + ", ".join(kv) + " }" if kv else "{}"))
            elif kind == 2:
                exprs.append(pick(["// a note", "/* block */", "/* two\nlines */"]))
            else:
                exprs.append("input -> aa -> bb -> output")
        return pick(["\n", "\n\n", " \n"]).join(exprs)

    accepted = 0
    for i in range(1500):
        if i % 4 == 3:
            n = rng.randint(1, 14)
            text = "".join(pick(vocab) + ("" if rng.rand() < 0.3 else " ") for _ in range(n))
        else:
            text = well_formed()
            for _ in range(rng.randint(0, 3) if rng.rand() < 0.5 else 0):     # a few token-level mutations
                at = rng.randint(len(text) + 1)
                text = text[:at] + pick(vocab) + text[at + rng.randint(0, 3):]
        try:
            ora = og.parse_config(text, True)
        except og.ConfigError:
            ora = None
        try:
            prod = rf.Config(text, True).nodes()
        except rf.RfError as e:
            assert e.status == 2, text
            prod = None
        assert (ora is None) == (prod is None), repr(text)
        if ora is not None:
            accepted += 1
            assert set(prod) == set(ora.graph_pipelines), repr(text)
            for name, node in prod.items():
                assert node["inputs"] == ora.graph_pipelines[name]["inputs"] and node["outputs"] == ora.graph_pipelines[name]["outputs"], repr(text)
                assert node["type"] == ora.type_of(name) and node["params"] == ora.params_of(name), repr(text)
    assert accepted > 50


def test_planner_matches_restatement_on_random_graphs():
    """300 generated graphs (chains, forks, in-place point ops): the C++ planner and the Python
    restatement agree on layers, aliases, allocated images and every resolved name; no plan lets
    a stencil write the image it reads; fused plans cover every node exactly once."""
    import numpy as np
    for seed in range(300):
        text = util.random_graph(np.random.RandomState(5000 + seed))
        p, layers, reuse = plans(text)
        assert p.layers() == layers and p.aliases() == reuse, text
        assert p.images() == og.GraphOracle(text, 4, 4, util.F32).allocated_images(), text
        for info in p.launch_info():
            if len(info["inputs"]) == 1 and info["radius"] > 0:
                assert p.resolve(info["inputs"][0]) != p.resolve(info["output"]), text
        fused = rf.Plan(rf.Config(text), 0)
        members = [m for info in fused.launch_info() for m in info["members"]]
        assert sorted(members) == sorted(rf.Config(text).nodes()), text
        for info in fused.launch_info():
            if info["radius"] > 0 and len(info["inputs"]) == 1:
                assert fused.resolve(info["inputs"][0]) != fused.resolve(info["output"]), text


IN_PLACE_BESIDE_A_READER = """
input -> n00 -> n01 -> mx:input_image0
n00 -> n02:image -> mx:input_image1
mx -> output
n00: gaussian5 { sigma: 1.54 }
n01: gaussian { sigma: 0.94, radius: 6 }
n02: colour_grade { slope: 0.54, offset: 0.066, saturation: 0.68 }
mx: combination { mix: 0.19 }
"""


def test_layer_with_an_in_place_writer_beside_a_reader_runs_in_order():
    """n01 and n02 share a layer and both read n00's image; n02 is written in place (n02:image) and
    is aliased onto that image (pipeline_graph.rs:400-411).  The reference dispatches the two
    behind one barrier (command.rs:194-240): a race.  The plan marks the layer serial -- plan
    order is name order, which is how the oracle executes it -- and leaves other layers alone
    (found by tests/test_gpu_fullsize.py::test_random_graphs_1080p_whole_frame)."""
    for flags in (0, NF):
        info = {l["label"]: l for l in rf.Plan(rf.Config(IN_PLACE_BESIDE_A_READER), flags).launch_info()}
        assert info["n02"]["output"] == info["n01"]["inputs"][0] == "n00:output_image"
        assert info["n01"]["serial"] and info["n02"]["serial"]
        assert not info["n00"]["serial"] and not info["mx"]["serial"]
    assert not any(l["serial"] for l in rf.Plan(rf.Config(util.DIAMOND), 0).launch_info())     # an ordinary fork/join stays concurrent


def test_in_place_head_with_other_consumers_is_not_fused():
    """`n00 -> n01:image -> n02 ...` beside `n00 -> n04`: n01 modifies n00's image in place and n04
    reads that image.  Fusing n01 into the chain behind it would leave n00's image untouched and n04
    would read the ungraded texels (found by scripts/fuzz_graphs.py): the in-place head keeps its
    own launch, the rest of the chain still fuses."""
    text = """input -> n00 -> n01:image -> n02 -> n03 -> mx:input_image0
n00 -> n04 -> mx:input_image1
mx -> output
n00: gaussian { sigma: 0.80, radius: 1 }
n01: colour_grade { slope: 1.39, offset: 0.098, saturation: 0.19 }
n02: gaussian9 { sigma: 2.0 }
n03: gaussian5 { sigma: 1.65 }
n04: passthrough {}
mx: combination { mix: 0.30 }"""
    info = {l["label"]: l for l in rf.Plan(rf.Config(text), 0).launch_info()}
    assert set(info) == {"n00", "n01", "n04", "n02+n03", "mx"}
    assert info["n01"]["output"] == info["n01"]["inputs"][0] == info["n04"]["inputs"][0] == "n00:output_image"
    assert info["n01"]["serial"] and info["n04"]["serial"]
    # with a single consumer the in-place node fuses like any other
    solo = "input -> n00 -> n01:image -> n02 -> output\nn00: sharpen {}\nn01: colour_grade {}\nn02: gaussian5 {}"
    assert rf.Plan(rf.Config(solo), rf.RF_GRAPH_NO_JIT).launches() in (["n00+n01", "n02"], ["n00", "n01+n02"])
    if rf.lib().rf_jit_available():
        assert rf.Plan(rf.Config(solo), 0).launches() == ["n00+n01+n02"]


def test_in_place_on_the_file_input_is_not_fused():
    """`input -> aa:image -> bb`: aa grades rf:file-input in place (pipeline_graph.rs:400-411 aliases
    onto the input image although it is never recycled).  The input persists from frame to frame, so
    the next frame sees graded texels -- in the reference and here; fusing aa away would change that."""
    text = "input -> aa:image -> bb -> output\naa: colour_grade { slope: 0.9 }\nbb: gaussian5 { sigma: 1.0 }"
    p = rf.Plan(rf.Config(text), 0)
    assert p.launches() == ["aa", "bb"] and p.resolve("aa:image") == "rf:file-input"
    assert rf.Plan(rf.Config(text.replace("aa:image", "aa")), 0).launches() == ["aa+bb"]


def test_graph_without_an_input_image_plans_without_crashing():
    """`sharpen -> output` with no `input` (allowed when no input file is given, config.rs:201): the node
    has no input image.  The planner (fusion pass included) must survive it."""
    for flags in (0, NF):
        p = rf.Plan(rf.Config("sharpen -> output", expects_input=False), flags)
        assert p.layers() == [["sharpen"]]
        assert p.launch_info() == []          # nothing launchable; rf_graph_create reports why (test_graph_errors)


def test_runs_of_in_place_nodes_fuse_only_when_nobody_else_sees_the_image():
    """The write of an in-place node lands on the allocation at the ROOT of its run of in-place nodes
    (aa:image -> bb:image both grade the image aa read).  Fusing is allowed only if every reader of
    that allocation is inside the group and it is not rf:file-input (scripts/fuzz_graphs.py,
    scripts/fuzz_strips.py found both shapes)."""
    # second in-place node of a run on the file input: still modifies the input every frame
    t = "input -> aa:image -> bb:image -> cc -> output\naa: colour_grade {}\nbb: colour_grade {}\ncc: sharpen {}"
    p = rf.Plan(rf.Config(t), 0)
    assert p.launches() == ["aa", "bb", "cc"] and p.resolve("bb:image") == "rf:file-input"
    # a run on an image that a second branch also reads (through its own in-place node)
    t = """input -> n00 -> n01:image -> n02:image -> n03 -> mx:input_image0
n00 -> n04:image -> mx:input_image1
mx -> output
n00: gaussian5 {}
n01: colour_grade {}
n02: colour_grade {}
n03: gaussian5 {}
n04: colour_grade {}
mx: combination { mix: 0.5 }"""
    assert sorted(rf.Plan(rf.Config(t), 0).launches()) == ["mx", "n00", "n01", "n02", "n03", "n04"]
    # the same run with no other reader fuses as before
    t = "input -> n00 -> n01:image -> n02:image -> output\nn00: gaussian5 {}\nn01: colour_grade {}\nn02: colour_grade {}"
    assert len(rf.Plan(rf.Config(t), rf.RF_GRAPH_NO_JIT).launches()) == 2
    if rf.lib().rf_jit_available():
        assert rf.Plan(rf.Config(t), 0).launches() == ["n00+n01+n02"]


def test_a_node_named_in_two_expressions_lists_its_output_once():
    """`n01 -> n02` ... `n02 -> n03`, `n02 -> n07`: config.rs:149-190 pushes n02's output once per
    occurrence.  With the duplicate, the aliasing pass (pipeline_graph.rs:398-424) remaps the output
    onto a free image and then, free list empty, ALSO records it as an allocation: an image that is
    alias and recyclable allocation at once, which later lands a stencil's output on its own input
    (scripts/fuzz_graphs.py with FUZZ_GEN=dag found it).  The planner works on sets."""
    import numpy as np
    text = util.random_dag(np.random.RandomState(500059))
    cfg = og.parse_config(text)
    assert cfg.graph_pipelines["n02"]["outputs"] == [("n02:output_image", "output_image")] * 2       # the config keeps both
    infos = og.synthesize(cfg)
    assert infos["n02"].output_images == [("n02:output_image", 1)]                                   # the planner does not
    for flags in (0, NF):
        p = rf.Plan(rf.Config(text), flags)
        info = p.launch_info()
        assert info, "the plan must be launchable"
        for l in info:
            if l["radius"] > 0 and len(l["inputs"]) == 1:
                assert l["inputs"][0] != l["output"]
        roots = set(p.images())
        assert all(p.resolve(a) in roots for a in p.aliases())          # every alias ends on an allocation


def test_planner_matches_restatement_on_random_dags():
    """400 graphs from the wide generator (several forks and joins, nodes read by many, in-place runs,
    type aliases): same layers, aliases and images as the restatement; every plan launchable; no
    stencil writes the image it reads."""
    import numpy as np
    for seed in range(400):
        text = util.random_dag(np.random.RandomState(9000 + seed))
        p, layers, reuse = plans(text)
        assert p.layers() == layers and p.aliases() == reuse, text
        for flags in (0, NF):
            q = rf.Plan(rf.Config(text), flags)
            info = q.launch_info()
            assert info, text
            assert sorted(m for l in info for m in l["members"]) == sorted(rf.Config(text).nodes()), text
            for l in info:
                if l["radius"] > 0 and len(l["inputs"]) == 1:
                    assert l["inputs"][0] != l["output"], text


# ---- storage buffers (SSBO edges) and `_rf_time` ---------------------------------------------------------------
SSBO = """input -> kw -> cv -> output
kw:ConvWeights -> cv:ConvWeights
kw: conv2d_weights { ksize: 5, sigma: 1.2 }
cv: conv2d { ksize: 5, sigma: 9.0 }"""


def test_planner_matches_restatement_on_random_dags_with_two_output_nodes():
    """random DAGs in which some nodes are `split_luma` (two output images, either or both wired): layers and aliasing against
    the restatement, and every launch that writes two images writes two DIFFERENT allocations"""
    import numpy as np
    n_split = 0
    for seed in range(200):
        text = util.random_dag(np.random.RandomState(7000 + seed), split=True)
        p, layers, reuse = plans(text)
        assert p.layers() == layers and p.aliases() == reuse, text
        for l in p.launch_info() + rf.Plan(rf.Config(text)).launch_info():
            assert len(set(l["outputs"])) == len(l["outputs"]) >= 1, text
            n_split += len(l["outputs"]) == 2
    assert n_split > 60


def test_storage_buffer_edges_follow_the_reference_rules():
    """A descriptor that is no image variable is looked up as a storage buffer by its block TYPE name
    (vkutils.rs:165-170, shader.rs:144-147); buffer edges order the layers like image edges
    (pipeline_graph.rs:438,:443); a buffer is sized to the largest block among its users (:158-175); an output on the
    binding of an input is the same buffer (:240-246).  Product plan == oracle restatement."""
    assert rf.lib().rf_registry_buffer_binding(b"conv2d", b"ConvWeights") == 3
    assert rf.lib().rf_registry_buffer_binding(b"conv2d_weights", b"ConvWeights") == 3
    assert rf.lib().rf_registry_buffer_binding(b"sharpen", b"ConvWeights") == -1
    p = rf.Plan(rf.Config(SSBO), NF)
    o = og.GraphOracle(SSBO, 8, 8, pixel.FMT_RGBA32F)
    assert p.layers() == o.layers == [["kw"], ["cv"]]
    assert p.buffers() == {"kw:ConvWeights": 3844} and sorted(o.ssbos) == ["kw:ConvWeights"] and o.ssbo_bytes["kw:ConvWeights"] == 3844
    # the buffer edge alone orders two nodes that share no image: cv2 must wait for kw although it reads only the input
    text = "input -> kw -> mx:input_image1\ninput -> cv2 -> mx:input_image0\nkw:ConvWeights -> cv2:ConvWeights\nmx -> output\n" \
           "kw: conv2d_weights { ksize: 3, sigma: 1.0 }\ncv2: conv2d { ksize: 3 }\nmx: combination { mix: 0.5 }"
    p = rf.Plan(rf.Config(text), NF)
    assert p.layers() == og.GraphOracle(text, 8, 8, pixel.FMT_RGBA32F).layers == [["kw"], ["cv2"], ["mx"]]
    # a conv2d that names a buffer nobody writes: "No buffer found for input" (pipeline_graph.rs:269)
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> cv -> output\ninput -> cv:ConvWeights\ncv: conv2d { ksize: 3 }"), NF).halo_schedule(True)
    assert "No buffer found for input rf:file-input" in str(e.value)
    # unknown descriptor: neither an image nor a buffer of the type
    with pytest.raises(rf.RfError) as e:
        rf.Plan(rf.Config("input -> kw -> output\nkw:Nope -> cv:Nope\ncv -> output\nkw: conv2d_weights {}\ncv: conv2d {}"), NF)
    assert "has no binding named: Nope" in str(e.value)
    # a node with a buffer edge is never fused into a chain
    fused = rf.Plan(rf.Config(SSBO.replace("-> cv -> output", "-> cv -> sh -> output") + "\nsh: sharpen {}"), 0).launches()
    assert "kw" in fused and not any("kw+" in l or "+kw" in l for l in fused)


def test_rf_time_member_exists_and_is_a_float():
    assert "pulse" in rf.registry_types() and "pulse" in og.NODE_TYPES
    assert og.NODE_TYPES["pulse"]["params"] == {"amount": "f32", "phase_rf_time": "f32"}
