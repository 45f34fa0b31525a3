"""Graph-level CPU oracle: config DSL -> layers -> image aliasing -> execution.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

A pure-Python restatement (small cases only) of
  * src/config/config_grammar.lalrpop:7-81 + src/config/config.rs:98-205 (parse)
  * src/vulkan/vkutils.rs:140-196                        (synthesize_config)
  * src/vulkan/pipeline_graph.rs:429-497                 (order_by_execution)
  * src/vulkan/pipeline_graph.rs:358-427                 (reusable_image_remapping)
  * src/vulkan/pipeline_graph.rs:205-237                 (image creation / binding)
  * src/vulkan/command.rs:166-242                        (one pass per node, barrier per layer)
  * src/render.rs:167-210                                (parameter initialisation)
with the per-pixel work done by rf_oracle.c (oracle/pixel.py).

Where the reference iterates a HashMap/HashSet (unspecified order) this
restatement iterates names in sorted order; results must not depend on it.
"""
import re

import numpy as np

from . import pixel
from . import user_stage

FILE_INPUT = "rf:file-input"      # pipeline_graph.rs:22
FINAL_OUTPUT = "rf:final-output"  # pipeline_graph.rs:23


class ConfigError(Exception):
    """The reference returns None + warnln!; the restatement raises."""


# --------------------------------------------------------------------------- #
# Lexer: LALRPOP's generated lexer = skip whitespace, longest match, literal
# terminals win ties over regex terminals (config_grammar.lalrpop:20-81).
# --------------------------------------------------------------------------- #
_TOKEN_RES = [
    ("LCOMMENT", re.compile(r"//[^\n\r]*[\n\r]*")),                                   # :24
    ("BCOMMENT", re.compile(r"/\*([^\*]*\*+[^\*/])*([^\*]*\*+|[^\*])*\*/")),          # :27
    ("DEC", re.compile(r"-?[0-9]+\.[0-9]+")),                                         # :76
    ("INT", re.compile(r"[0-9]+")),                                                   # :75
    ("STR", re.compile(r"[a-zA-Z_][a-zA-Z0-9_-]+")),                                  # :81
]
_LITERALS = ["->", "{}", "{", "}", ":", ",", "true", "false"]


def _lex(text):
    toks, i, n = [], 0, len(text)
    while i < n:
        if text[i].isspace():
            i += 1
            continue
        best_len, best_kind = 0, None
        for lit in _LITERALS:
            if text.startswith(lit, i) and len(lit) > best_len:
                best_len, best_kind = len(lit), lit
        for kind, rx in _TOKEN_RES:
            m = rx.match(text, i)
            if m and m.end() - i > best_len:      # strictly longer: literals win ties
                best_len, best_kind = m.end() - i, kind
        # The block-comment regex accepts EVERY string that starts with "/*" and ends
        # with "*/" (its second group generates any text), and the lexer takes the
        # longest match: a block comment runs to the LAST "*/" of the file.
        if text.startswith("/*", i):
            j = text.rfind("*/")
            if j >= i + 2 and j + 2 - i > best_len:
                best_len, best_kind = j + 2 - i, "BCOMMENT"
        if best_kind is None:
            raise ConfigError("Invalid token %r at offset %d" % (text[i], i))
        toks.append((best_kind, text[i:i + best_len]))
        i += best_len
    return toks


def _parse_exprs(toks, syntax=False):
    """ExprList (config_grammar.lalrpop:7-28): returns [('graph', members) | ('pipeline', name, type, params)];
    syntax: comments are kept as ('comment', text) and a pipeline's parameters are the [key, value] pairs in source order."""
    out, i, n = [], 0, len(toks)

    def need(kind, at):
        if at >= n or toks[at][0] != kind:
            got = toks[at][1] if at < n else "<eof>"
            raise ConfigError("Unrecognized token %r, expected %s" % (got, kind))
        return toks[at][1]

    if n == 0:
        raise ConfigError("no expressions")
    while i < n:
        kind, val = toks[i]
        if kind in ("LCOMMENT", "BCOMMENT"):
            if syntax:
                out.append(("comment", val))
            i += 1
            continue
        name = need("STR", i)
        i += 1
        desc = None
        if i < n and toks[i][0] == ":":
            second = need("STR", i + 1)
            i += 2
            if i < n and toks[i][0] in ("{", "{}"):
                # PipelineField  name : type { k: v, ... } | {}     (:44-51)
                params, fields = {}, []
                if toks[i][0] == "{}":
                    i += 1
                else:
                    i += 1
                    while True:
                        key = need("STR", i)
                        need(":", i + 1)
                        if i + 2 >= n or toks[i + 2][0] not in ("INT", "DEC", "true", "false"):
                            raise ConfigError("bad parameter value")
                        params[key] = toks[i + 2][1]       # HashMap insert: last one wins
                        fields.append([key, toks[i + 2][1]])
                        i += 3
                        if i < n and toks[i][0] == ",":
                            i += 1
                            continue
                        need("}", i)
                        i += 1
                        break
                out.append(("pipeline", name, second, fields if syntax else params))
                continue
            desc = second
        # GraphExpr: at least two members joined by "->"                  (:30-42)
        members = [(name, desc)]
        need("->", i)
        while i < n and toks[i][0] == "->":
            mname = need("STR", i + 1)
            i += 2
            mdesc = None
            if i < n and toks[i][0] == ":":
                mdesc = need("STR", i + 1)
                i += 2
            members.append((mname, mdesc))
        out.append(("graph", members))
    return out


def parse_syntax(text):
    """The generated parser alone (config.rs:105): {"exprs": [["pipeline", name, type, [[key, value], ...]] |
    ["graph", [[name, descriptor | None], ...]] | ["comment", text]]}; ConfigError for a text the grammar rejects.
    tests/test_grammar_fixtures.py holds it to vectors derived from the reference's grammar file."""
    exprs = []
    for e in _parse_exprs(_lex(text), syntax=True):
        if e[0] == "pipeline":
            exprs.append(["pipeline", e[1], e[2], e[3]])
        elif e[0] == "graph":
            exprs.append(["graph", [[n, d] for n, d in e[1]]])
        else:
            exprs.append(["comment", e[1]])
    return {"exprs": exprs}


class Config:
    """config.rs:17-38.  graph_pipelines: name -> {'inputs': [(resource, descriptor)], 'outputs': [...]};
    pipeline_instances: name -> (type, {param: string})."""

    def __init__(self):
        self.graph_pipelines = {}
        self.pipeline_instances = {}

    def type_of(self, node):
        # config.rs:59-75 (add_file_paths): instance type, else the node name
        inst = self.pipeline_instances.get(node)
        return inst[0] if inst else node

    def params_of(self, node):
        inst = self.pipeline_instances.get(node)
        return dict(inst[1]) if inst else {}


def parse_config(text, expects_input=True):
    """config.rs:98-205."""
    if not text.strip():
        raise ConfigError("Empty configuration given to parse")            # :99-102
    exprs = _parse_exprs(_lex(text))
    cfg = Config()
    found_input = found_output = False
    for e in exprs:
        if e[0] == "pipeline":
            cfg.pipeline_instances[e[1]] = (e[2], e[3])                   # :191-195
            continue
        graph = e[1]
        for i, (name, desc) in enumerate(graph):                          # :149-190
            if name == "input":
                found_input = True
                continue
            if name == "output":
                found_output = True
                continue
            info = cfg.graph_pipelines.setdefault(name, {"inputs": [], "outputs": []})
            if i > 0:
                prev_name, prev_desc = graph[i - 1]
                descriptor = desc if desc is not None else "input_image"
                resource = FILE_INPUT if prev_name == "input" else \
                    "%s:%s" % (prev_name, prev_desc if prev_desc is not None else "output_image")
                info["inputs"].append((resource, descriptor))
            if i + 1 < len(graph):
                next_name = graph[i + 1][0]
                descriptor = desc if desc is not None else "output_image"
                resource = FINAL_OUTPUT if next_name == "output" else "%s:%s" % (name, descriptor)
                info["outputs"].append((resource, descriptor))
    if not cfg.graph_pipelines:
        raise ConfigError("Configuration had an empty graph")              # :200
    if found_input and not expects_input:
        raise ConfigError("Found 'input' in pipeline configuration but no input image was specified")  # :201
    if not found_output:
        raise ConfigError("'output' is never used in the pipeline configuration")                      # :202
    return cfg


# --------------------------------------------------------------------------- #
# Node-type table: what SPIR-V reflection gives the reference
# (shader.rs:106-160): image variable name -> binding index, uniform members.
# --------------------------------------------------------------------------- #
_IO = {"input_image": 0, "output_image": 1}
_IO_RW = {"input_image": 0, "output_image": 1, "image": 2}

def _gauss_params(radius, with_radius=False):
    """sigma (+ radius) and the optional explicit weights w0 .. wR (shaders/gaussian*.comp)."""
    p = {"sigma": "f32"}
    if with_radius:
        p["radius"] = "i32"
    p.update({"w%d" % i: "f32" for i in range(radius + 1)})
    return p


NODE_TYPES = {
    "passthrough":  {"images": _IO,    "params": {}},
    "gaussian5":    {"images": _IO,    "params": _gauss_params(2)},
    "gaussian9":    {"images": _IO,    "params": _gauss_params(4)},
    "gaussian":     {"images": _IO,    "params": _gauss_params(15, True)},
    "colour_grade": {"images": _IO_RW, "params": {"slope": "f32", "offset": "f32", "saturation": "f32"}},
    "sharpen":      {"images": _IO,    "params": {"amount": "f32"}},
    # storage buffers are found by the block TYPE name (shader.rs:144-147): "buffers" = name -> (binding, bytes)
    "conv2d":       {"images": _IO,    "params": {"ksize": "i32", "sigma": "f32"}, "buffers": {"ConvWeights": (3, 961 * 4)}},
    "conv2d_weights": {"images": _IO,  "params": {"ksize": "i32", "sigma": "f32"}, "buffers": {"ConvWeights": (3, 961 * 4)}},
    # `phase_rf_time` receives the seconds since start every frame (render.rs:212-223)
    "pulse":        {"images": _IO,    "params": {"amount": "f32", "phase_rf_time": "f32"}},
    "combination":  {"images": {"input_image0": 0, "input_image1": 1, "output_image": 2},
                     "params": {"mix": "f32"}},
}
# a node with TWO output images (one image per output binding, pipeline_graph.rs:205-224)
NODE_TYPES["split_luma"] = {"images": {"input_image": 0, "luma_image": 1, "chroma_image": 2}, "params": {}}
NODE_TYPES["colour-grade"] = NODE_TYPES["colour_grade"]
NODE_TYPES["colour_grade_inplace"] = {"images": {"image": 0}, "params": NODE_TYPES["colour_grade"]["params"]}
NODE_TYPES["grade"] = NODE_TYPES["colour_grade"]


def register_user_type(name, path):
    """a filter type that is a file, {shader_path}/{name}.stage.hip: reflected and compiled for the host by
    oracle/user_stage.py.  Returns the UserType.  Test infrastructure: lets generated graphs hold user types."""
    ut = user_stage.UserType(name, path)
    NODE_TYPES[name] = ut.node_type()
    return ut


def _parse_param(s, ty):
    """render.rs:169-185: Rust str::parse::<f32|i32|bool>, failure -> 0 + warning."""
    if s is None:
        return 0          # absent parameter: zero-filled (render.rs:200-203)
    try:
        if ty == "f32":
            return float(np.float32(float(s)))
        if ty == "i32":
            if not re.fullmatch(r"[+-]?[0-9]+", s):
                raise ValueError
            return int(s)
        if ty == "bool":
            return {"true": True, "false": False}[s]
    except (ValueError, KeyError):
        return 0
    return 0


class PipelineInfo:
    """pipeline.rs:17-25: [(resource_name, binding)] for images and storage buffers"""

    def __init__(self, name, type_name):
        self.name, self.type = name, type_name
        self.input_images, self.output_images = [], []
        self.input_ssbos, self.output_ssbos = [], []
        self.params = {}


def synthesize(cfg):
    """vkutils.rs:140-196."""
    infos = {}
    for name in sorted(cfg.graph_pipelines):
        tname = cfg.type_of(name)
        t = NODE_TYPES.get(tname)
        if t is None:
            raise ConfigError("no node type %r" % tname)      # Shader::from_path -> None
        info = PipelineInfo(name, tname)
        for key, dst, bdst in (("inputs", info.input_images, info.input_ssbos), ("outputs", info.output_images, info.output_ssbos)):
            for resource, descriptor in cfg.graph_pipelines[name][key]:
                if descriptor not in t["images"]:
                    # not an image variable: a storage buffer by its block type name (vkutils.rs:165-170)
                    if descriptor in t.get("buffers", {}):
                        if (resource, t["buffers"][descriptor][0]) not in bdst:
                            bdst.append((resource, t["buffers"][descriptor][0]))
                        continue
                    raise ConfigError("Shader has no binding named: %s" % descriptor)   # :179
                # a node named in several graph expressions repeats the same (resource, binding)
                # (config.rs:149-190 pushes per occurrence); the planner works on sets -- with the
                # duplicate, the aliasing pass first remaps the output and then also allocates it,
                # and that alias-and-allocation later lands a stencil's output on its own input
                if (resource, t["images"][descriptor]) not in dst:
                    dst.append((resource, t["images"][descriptor]))
        given = cfg.params_of(name)
        info.params = {k: _parse_param(given.get(k), ty) for k, ty in t["params"].items()}
        infos[name] = info
    return infos


def order_by_execution(infos):
    """pipeline_graph.rs:429-497: list of layers (each a name-sorted list)."""
    unexecuted = set(infos)

    def input_nodes(info):
        ins = [r for r, _ in info.input_images]
        bins = [r for r, _ in info.input_ssbos]           # storage-buffer edges order nodes like image edges (:438,:443)
        return [c for c, ci in infos.items() if any(r in ins for r, _ in ci.output_images) or any(r in bins for r, _ in ci.output_ssbos)]

    layers = []
    while unexecuted:
        snapshot = sorted(unexecuted)
        layer = []
        for node in snapshot:
            if not any(n in snapshot for n in input_nodes(infos[node])):
                unexecuted.discard(node)
                layer.append(node)
        if len(snapshot) == len(unexecuted):
            raise ConfigError("Graph incorrectly constructed. Failed to add nodes into execution: %s" % snapshot)
        layers.append(layer)
    return layers


def _remap(name, mapping):
    # pipeline_graph.rs:75-79
    while name in mapping:
        name = mapping[name]
    return name


def reusable_image_remapping(layers, infos, literal=False):
    """pipeline_graph.rs:358-427, with one deliberate difference: "is this allocation still
    used through a remap" follows the alias chain to its end.  The reference tests ONE level
    (images_have_remap, :364-369: `image_reuse.get(image_name) == name`), so an image read
    through a two-level alias -- a point op written in place (X:image) on an image that was
    itself a recycled allocation -- counts as free while a later node still reads it; the next
    output then lands on the image its own node reads and a stencil runs in place (undefined
    output in the reference).  Following the chain only ever keeps an image allocated longer:
    every plan the reference gets right is unchanged.  literal=True gives the reference's own
    result (tests/test_config_plan.py shows a graph where the two differ)."""
    free_images, images, reuse = [], [], {}

    def has_remap(name, imgs):
        if literal:
            return any(reuse.get(img) == name for img, _ in imgs)
        return any(img in reuse and _remap(img, reuse) == name for img, _ in imgs)

    def node_uses(info, name):
        return (any(n == name for n, _ in info.input_images) or
                any(n == name for n, _ in info.output_images) or
                has_remap(name, info.input_images) or has_remap(name, info.output_images))

    def still_in_use(name, start_layer):
        return any(node_uses(infos[n], name) for layer in layers[start_layer:] for n in layer)

    for li, layer in enumerate(layers):
        for name in sorted(images):
            if name in free_images:
                continue
            if not still_in_use(name, li):
                free_images.append(name)
        for node in layer:
            info = infos[node]
            for image_name, out_binding in info.output_images:
                point_op = False
                for input_name, in_binding in info.input_images:
                    if out_binding == in_binding:
                        point_op = True
                        reuse[image_name] = input_name
                if point_op:
                    continue
                if not free_images:
                    if image_name not in images:
                        images.append(image_name)
                else:
                    reuse[image_name] = free_images.pop()
    return reuse


class GraphOracle:
    """Builds and runs a graph on numpy images, honouring the aliasing plan so
    that an unsafe alias would corrupt the result exactly as it would on device."""

    def __init__(self, text, W, H, fmt, expects_input=True):
        self.cfg = parse_config(text, expects_input)
        self.infos = synthesize(self.cfg)
        self.layers = order_by_execution(self.infos)
        self.reuse = reusable_image_remapping(self.layers, self.infos)
        self.W, self.H, self.fmt = W, H, fmt
        self.images = {}
        self.buffers = {}     # node -> conv weights override
        # storage buffers (pipeline_graph.rs:142-175, :240-265): size = max over users of the block's bytes; an output on
        # the binding of an input is the same buffer (point op); one zero-filled buffer per resolved name
        self.ssbo_remap, self.ssbo_bytes, self.ssbos = {}, {}, {}
        for layer in self.layers:
            for node in layer:
                info = self.infos[node]
                size = max([b[1] for b in NODE_TYPES[info.type].get("buffers", {}).values()] or [0])
                for r, _ in info.input_ssbos + info.output_ssbos:
                    self.ssbo_bytes[r] = max(self.ssbo_bytes.get(r, 0), size)
                for ro, bo in info.output_ssbos:
                    for ri, bi in info.input_ssbos:
                        if bo == bi:
                            self.ssbo_remap[ro] = ri
        for layer in self.layers:
            for node in layer:
                for r, _ in self.infos[node].output_ssbos:
                    nm = _remap(r, self.ssbo_remap)
                    self.ssbos.setdefault(nm, np.zeros(self.ssbo_bytes[nm] // 4, np.float32))
        # pipeline_graph.rs:205-224: FILE_INPUT once, outputs through the remap
        for layer in self.layers:
            for node in layer:
                info = self.infos[node]
                for r, _ in info.input_images:
                    if r == FILE_INPUT and r not in self.images:
                        self.images[r] = pixel.new_image(W, H, fmt)
                for r, _ in info.output_images:
                    nm = _remap(r, self.reuse)
                    if nm not in self.images:
                        self.images[nm] = pixel.new_image(W, H, fmt)

    # -- plan views used by the parity tests ---------------------------------
    def allocated_images(self):
        return sorted(self.images)

    def image_of(self, resource):
        return _remap(resource, self.reuse)

    def set_param(self, node, key, value):
        self.infos[node].params[key] = value

    def set_time(self, seconds):
        """update_ubos, render.rs:212-223: every member whose name ends in `_rf_time`."""
        for info in self.infos.values():
            for k in info.params:
                if k.endswith("_rf_time"):
                    info.params[k] = float(np.float32(seconds))

    def set_weights(self, node, weights):
        self.buffers[node] = np.ascontiguousarray(weights, np.float32)

    def upload_raw(self, img):
        assert img.shape == (self.H, self.W, 4) and img.dtype == pixel.dtype_of(self.fmt)
        self.images[FILE_INPUT][...] = img

    def upload_srgb8(self, rgba):
        self.images[FILE_INPUT][...] = pixel.upload_srgb8(rgba, self.fmt)

    def _run_node(self, info):
        def img(resource):
            nm = _remap(resource, self.reuse)
            if nm not in self.images:
                raise ConfigError("No image found for input %s" % nm)       # pipeline_graph.rs:236
            return self.images[nm]

        if not info.output_images:
            return                       # nothing observable
        for r, _ in info.input_ssbos:
            if _remap(r, self.ssbo_remap) not in self.ssbos:
                raise ConfigError("No buffer found for input %s" % r)       # pipeline_graph.rs:269
        if not info.input_images:
            raise ConfigError("node %s has no input image" % info.name)
        t, p = info.type, info.params
        if "user" in NODE_TYPES.get(t, {}):
            # a type that is a FILE (register_user_type): the file's own code, compiled for the host (oracle/user_stage.py)
            ut = NODE_TYPES[t]["user"]
            in_by = {b: r for r, b in info.input_images}
            out_by = {b: r for r, b in info.output_images}
            for i, nm in enumerate(ut.inputs):
                if i not in in_by:
                    raise ConfigError("node %s needs an image wired to %s" % (info.name, nm))
            srcs = [img(in_by[i]) for i in range(len(ut.inputs))]
            dsts = [img(out_by[b]) if b in out_by else None for b in ut.out_binding]
            if ut.radius > 0 and any(d is not None and any(d is s_ for s_ in srcs) for d in dsts):
                raise ConfigError("in-place execution of a stencil node")
            buf_in = buf_out = None
            if ut.buf_in:
                if not info.input_ssbos:
                    raise ConfigError("node %s needs a storage buffer wired to %s" % (info.name, ut.buf_in[0]))
                buf_in = self.ssbos[_remap(info.input_ssbos[0][0], self.ssbo_remap)]
            if ut.buf_out and info.output_ssbos:
                buf_out = self.ssbos[_remap(info.output_ssbos[0][0], self.ssbo_remap)]
            user_stage.run(ut, p, srcs, dsts, buf_in, buf_out)
            return
        if t == "split_luma":
            by_binding = {b: r for r, b in info.output_images}
            pixel.split_luma(img(info.input_images[0][0]), img(by_binding[1]) if 1 in by_binding else None, img(by_binding[2]) if 2 in by_binding else None)
            return
        dst = img(info.output_images[0][0])
        if t == "combination":
            by_binding = {b: r for r, b in info.input_images}
            if 0 not in by_binding or 1 not in by_binding:
                raise ConfigError("combination needs input_image0 and input_image1")
            pixel.mix(img(by_binding[0]), img(by_binding[1]), p["mix"], dst=dst)
            return
        src = img(info.input_images[0][0])
        if src is dst and t not in ("colour_grade", "colour-grade", "grade", "colour_grade_inplace", "passthrough", "conv2d_weights", "pulse"):
            raise ConfigError("in-place execution of a stencil node")
        if t == "passthrough":
            if src is not dst:
                pixel.passthrough(src, dst)
        elif t in ("gaussian5", "gaussian9", "gaussian"):
            radius = {"gaussian5": 2, "gaussian9": 4}.get(t)
            if radius is None:
                radius = min(max(int(p["radius"]), 0), pixel.MAX_RADIUS)
            given = [p["w%d" % i] for i in range(radius + 1)]
            if any(w != 0 for w in given):      # explicit weights replace the derived kernel
                pixel.gaussian(src, radius, weights=np.asarray(given, np.float32), dst=dst)
            else:
                pixel.gaussian(src, radius, sigma=p["sigma"], dst=dst)
        elif t in ("colour_grade", "colour-grade", "grade", "colour_grade_inplace"):
            pixel.colour_grade(src, p["slope"], p["offset"], p["saturation"], dst=dst)
        elif t == "sharpen":
            pixel.sharpen(src, p["amount"], dst=dst)
        elif t == "conv2d":
            w = self.buffers.get(info.name)
            if info.input_ssbos:
                # weights through a storage-buffer edge: the first K*K floats of the wired buffer, whoever wrote them
                K = conv_ksize(p["ksize"])
                w = self.ssbos[_remap(info.input_ssbos[0][0], self.ssbo_remap)][:K * K].reshape(K, K).copy()
            elif w is None:
                w = default_conv_weights(p["ksize"], p["sigma"])
            pixel.conv2d(src, w, dst=dst)
        elif t == "conv2d_weights":
            # writes the K x K weights (host-derived defaults, or what set_weights gave this node) and passes its image through
            w = self.buffers.get(info.name)
            if w is None:
                w = default_conv_weights(p["ksize"], p["sigma"])
            for r, _ in info.output_ssbos:
                buf = self.ssbos[_remap(r, self.ssbo_remap)]
                buf[:] = 0
                buf[:w.size] = np.ascontiguousarray(w, np.float32).reshape(-1)
            if src is not dst:
                pixel.passthrough(src, dst)
        elif t == "pulse":
            # a colour grade whose slope breathes with time: slope = fmaf(amount, frac(phase_rf_time), 1), offset 0, saturation 1
            pixel.colour_grade(src, pixel.pulse_slope(p["amount"], p["phase_rf_time"]), 0.0, 1.0, dst=dst)
        else:
            raise ConfigError("no node type %r" % t)

    def execute(self):
        # command.rs:220-240: layer by layer; inside a layer order is unspecified
        for layer in self.layers:
            for node in layer:
                self._run_node(self.infos[node])

    def output(self):
        # pipeline_graph.rs:86-96: the output may itself be an alias
        return self.images[_remap(FINAL_OUTPUT, self.reuse)]

    def download_raw(self):
        return self.output().copy()

    def download_srgb8(self):
        return pixel.download_srgb8(self.output())


def conv_ksize(ksize):
    """conv2d 'ksize' parameter -> odd K in [1, 31] (authored rule)."""
    k = int(ksize)
    if k < 1:
        k = 1
    if k > 2 * pixel.MAX_RADIUS + 1:
        k = 2 * pixel.MAX_RADIUS + 1
    if k % 2 == 0:
        k -= 1
    return k


def default_conv_weights(ksize, sigma):
    """Default conv2d weights: outer product of the normalised 1-D gaussian in
    double, rounded once to f32 (authored rule)."""
    import math
    K = conv_ksize(ksize)
    r = K // 2
    if not (sigma > 0.0):
        g = [1.0] + [0.0] * r
    else:
        s2 = 2.0 * float(np.float32(sigma)) ** 2
        e = [math.exp(-(i * i) / s2) for i in range(r + 1)]
        tot = e[0]
        for i in range(1, r + 1):
            tot += 2.0 * e[i]
        g = [v / tot for v in e]
    w = np.zeros((K, K), dtype=np.float32)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            w[dy + r, dx + r] = np.float32(g[abs(dy)] * g[abs(dx)])
    return w
