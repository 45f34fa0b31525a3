"""Shared helpers of the parity tests: run the same config text through the CPU oracle
and through librfhip.so (C ABI), return the raw output texels."""
import os

import numpy as np

import reforge_amd as rf
from oracle import graph as ograph
from oracle import pixel

F32 = rf.RF_FORMAT_RGBA32F
U8 = rf.RF_FORMAT_RGBA8

CHAIN3 = """
input -> blur -> grade -> sharp -> output
blur:  gaussian5    { sigma: 1.0 }
grade: colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp: sharpen      { amount: 0.5 }
"""

# the 5-stage chain of BASELINE config 4
CHAIN5 = """
input -> blur -> grade -> sharp -> wide -> finish -> output
blur:   gaussian5    { sigma: 1.0 }
grade:  colour_grade { slope: 1.1, offset: -0.02, saturation: 1.2 }
sharp:  sharpen      { amount: 0.5 }
wide:   gaussian9    { sigma: 2.0 }
finish: colour_grade { slope: 0.95, offset: 0.01, saturation: 0.9 }
"""

# same shape with a gaussian5 in the middle: fuses as 3 + 2 launches (two fused launches per frame)
CHAIN5_SPLIT = CHAIN5.replace("gaussian9    { sigma: 2.0 }", "gaussian5    { sigma: 2.0 }")

# a node with TWO output images (split_luma: luma_image, chroma_image), each processed on its own and joined again
SPLIT2 = """
input -> sp
sp:luma_image -> lg -> mx:input_image0
sp:chroma_image -> cg -> mx:input_image1
mx -> output
sp: split_luma {}
lg: gaussian5 { sigma: 1.0 }
cg: colour_grade { slope: 1.2, offset: -0.05, saturation: 1.1 }
mx: combination { mix: 0.4 }
"""

DIAMOND = """
input -> blur -> mixer:input_image0
input -> sharp -> mixer:input_image1
mixer -> output
blur:  gaussian5   { sigma: 1.5 }
sharp: sharpen     { amount: 0.75 }
mixer: combination { mix: 0.25 }
"""


def run_oracle(text, img, weights=None):
    H, W, _ = img.shape
    g = ograph.GraphOracle(text, W, H, pixel.fmt_of(img))
    for node, w in (weights or {}).items():
        g.set_weights(node, w)
    g.upload_raw(img)
    g.execute()
    return g.download_raw()


def run_hip(ctx, text, img, flags=0, weights=None, rows_per_chunk=None, num_frames=1, slot=0, conv_path=0, exec_flags=0, texels_per_lane=0):
    H, W, _ = img.shape
    g = rf.Graph(ctx, rf.Config(text), W, H, pixel.fmt_of(img), num_frames=num_frames, flags=flags,
                 rows_per_chunk=rows_per_chunk or 0, conv_path=conv_path, exec_flags=exec_flags, texels_per_lane=texels_per_lane)
    try:
        for node, w in (weights or {}).items():
            g.set_weights(node, w)
        g.upload_raw(img)
        g.execute(slot)
        g.wait(slot)
        return g.download_raw(slot)
    finally:
        g.close()


def assert_same(got, want, what=""):
    """Bit-exact for both formats (the bar is pixel-exact rgba8 / <= 1 ulp rgba32f; the
    kernels are built to be bit-identical, so the tests hold them to 0 ulp)."""
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    if got.tobytes() == want.tobytes():
        return
    if got.dtype == np.float32:
        a, b = got.view(np.uint32).astype(np.int64), want.view(np.uint32).astype(np.int64)
    else:
        a, b = got.astype(np.int64), want.astype(np.int64)
    bad = np.argwhere(a != b)
    y, x, c = bad[0]
    raise AssertionError("%s: %d of %d values differ; first at (y=%d,x=%d,c=%d): got %r want %r; max |diff| = %d (ulp or codes)" % (
        what, len(bad), a.size, y, x, c, got[y, x, c], want[y, x, c], np.abs(a - b).max()))


def synthetic(W, H, fmt, seed=0x5EED0002):
    return pixel.fill_synthetic(W, H, fmt, seed)


def random_graph(rng):
    """A random pipeline config: a chain of 1..6 nodes (optionally one fork/join through a
    `combination`), node types and parameters drawn at random, some point ops written in place
    (`name:image`).  Names are unique, two characters or more (config_grammar.lalrpop:81)."""
    kinds = ["passthrough", "gaussian5", "gaussian9", "gaussian", "colour_grade", "grade_inplace", "sharpen", "conv2d"]
    decl, names = [], []

    def node(i):
        kind = kinds[rng.randint(len(kinds))]
        name = "n%02d" % i
        if kind == "passthrough":
            decl.append("%s: passthrough {}" % name)
        elif kind in ("gaussian5", "gaussian9"):
            decl.append("%s: %s { sigma: %.2f }" % (name, kind, rng.uniform(0.4, 3.0)))
        elif kind == "gaussian":
            decl.append("%s: gaussian { sigma: %.2f, radius: %d }" % (name, rng.uniform(0.5, 4.0), rng.randint(0, 8)))
        elif kind in ("colour_grade", "grade_inplace"):
            decl.append("%s: colour_grade { slope: %.2f, offset: %.3f, saturation: %.2f }" % (name, rng.uniform(0.5, 1.5), rng.uniform(-0.1, 0.1), rng.uniform(0.0, 2.0)))
        elif kind == "sharpen":
            decl.append("%s: sharpen { amount: %.2f }" % (name, rng.uniform(0.0, 1.5)))
        else:
            decl.append("%s: conv2d { ksize: %d, sigma: %.2f }" % (name, (3, 5, 9)[rng.randint(3)], rng.uniform(0.6, 2.0)))
        return name + (":image" if kind == "grade_inplace" else "")

    n = rng.randint(1, 7)
    chain = [node(i) for i in range(n)]
    lines = []
    if n >= 3 and rng.rand() < 0.35:
        # fork after the first node, two branches, joined by a combination
        cut = rng.randint(1, n - 1)
        left, right = chain[1:cut + 1], chain[cut + 1:]
        head = chain[0].split(":")[0]            # the forked image must be materialised: no in-place head
        lines.append("input -> %s" % " -> ".join([head] + left + ["mx:input_image0"]))
        lines.append("%s -> %s" % (head, " -> ".join(right + ["mx:input_image1"])))
        lines.append("mx -> output")
        decl.append("mx: combination { mix: %.2f }" % rng.uniform(0.0, 1.0))
    else:
        lines.append("input -> %s -> output" % " -> ".join(chain))
    return "\n".join(lines + decl)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHADERS = os.path.join(ROOT, "shaders")
USER_TYPES = ("invert", "edge_detect", "unsharp_mask", "tone_curve", "apply_curve", "local_contrast", "streak")


def register_user_types():
    """the shipped stage files ({shader_path} = shaders/) on BOTH sides: the library looks them up there, the oracle compiles the
    same files for the host (oracle/user_stage.py).  Returns the previous shader path of the library (restore it afterwards)."""
    for n in USER_TYPES:
        if "user" not in ograph.NODE_TYPES.get(n, {}):
            ograph.register_user_type(n, os.path.join(SHADERS, n + ".stage.hip"))
    old = rf.shader_path()
    rf.set_shader_path(SHADERS)
    return old


def random_dag(rng, split=False, user=False):
    """(split=True: some nodes are `split_luma`, a node with TWO output images, each of which later nodes may read.
    user=True: some nodes are USER types -- shaders/*.stage.hip: invert and edge_detect (row stages that fuse), unsharp_mask (two
    inputs, up to two outputs read), tone_curve -> apply_curve (a storage-buffer edge), local_contrast (RADIUS 2, read through a window);
    call register_user_types() first.)
    A wider generator than random_graph: up to 9 nodes, any earlier node's output (or the input)
    may feed a new node, `combination` joins appear anywhere, type aliases, large radii and kernels,
    in-place point ops anywhere.  The last node drives the output; dangling nodes are pruned by
    making every node reachable from it (a node nobody reads feeds a final join)."""
    point = ["colour_grade", "grade", "colour-grade"]
    decl, edges = [], []            # edges: list of chains as token lists
    outputs = ["input"]             # names whose output image can be read
    n = rng.randint(1, 10)
    readers = {}
    for i in range(n):
        name = "n%02d" % i
        roll = rng.rand()
        src = outputs[rng.randint(len(outputs))] if rng.rand() < 0.35 else outputs[-1]
        if split and rng.rand() < 0.18:
            # one input, two readable outputs; a later node (or the final joins) must read each wired one
            sname = "s%02d" % i
            decl.append("%s: split_luma {}" % sname)
            edges.append([src, sname])
            readers[src] = readers.get(src, 0) + 1
            which = rng.randint(3)               # 0: both outputs offered, 1: luma only, 2: chroma only
            if which != 2:
                outputs.append(sname + ":luma_image")
            if which != 1:
                outputs.append(sname + ":chroma_image")
            continue
        if user and rng.rand() < 0.3:
            kind = ["invert", "edge_detect", "unsharp_mask", "curve", "local_contrast", "streak"][rng.randint(6)]
            if kind == "invert":
                decl.append("%s: invert { enabled: %s, strength: %.2f }" % (name, "true" if rng.rand() < 0.8 else "false", rng.uniform(0.0, 1.0)))
            elif kind == "edge_detect":
                decl.append("%s: edge_detect { scale: %.2f }" % (name, rng.uniform(0.2, 3.0)))
            elif kind in ("local_contrast", "streak"):          # RADIUS 2 / 7: a node of its own that reads its input through a window (LDS tiles)
                decl.append("%s: %s { amount: %.2f }" % (name, kind, rng.uniform(0.0, 1.5)))
            elif kind == "unsharp_mask" and len(outputs) >= 2:
                a, b = [outputs[k] for k in rng.choice(len(outputs), 2, replace=False)]
                decl.append("%s: unsharp_mask { amount: %.2f, threshold: %.3f }" % (name, rng.uniform(0.0, 2.0), rng.uniform(0.0, 0.1)))
                edges.append([a, name + ":input_image"])
                edges.append([b, name + ":blurred_image"])
                readers[a] = readers.get(a, 0) + 1
                readers[b] = readers.get(b, 0) + 1
                which = rng.randint(3)               # 0: both outputs offered to later nodes, 1: output_image only, 2: mask_image only
                if which != 2:
                    outputs.append(name)
                if which != 1:
                    outputs.append(name + ":mask_image")
                continue
            elif kind == "curve":
                # a node that fills a storage buffer, its reader behind it in the image chain AND wired to the buffer by its block type name
                tc = "t%02d" % i
                decl.append("%s: tone_curve { gamma: %.2f, lift: %.3f }" % (tc, rng.uniform(-0.8, 0.8), rng.uniform(0.0, 0.2)))
                decl.append("%s: apply_curve { strength: %.2f }" % (name, rng.uniform(0.0, 1.0)))
                edges.append([src, tc, name])
                edges.append([tc + ":ToneCurve", name + ":ToneCurve"])
                readers[src] = readers.get(src, 0) + 1
                outputs.append(name)
                continue
            else:
                decl.append("%s: invert { enabled: true, strength: 0.5 }" % name)
            edges.append([src, name])
            readers[src] = readers.get(src, 0) + 1
            outputs.append(name)
            continue
        if roll < 0.15 and len(outputs) >= 3:
            a, b = [outputs[k] for k in rng.choice(len(outputs), 2, replace=False)]
            if a != "input" or b != "input":
                decl.append("%s: combination { mix: %.2f }" % (name, rng.uniform(0, 1)))
                edges.append([a, name + ":input_image0"])
                edges.append([b, name + ":input_image1"])
                readers[a] = readers.get(a, 0) + 1
                readers[b] = readers.get(b, 0) + 1
                outputs.append(name)
                continue
        kind = ["passthrough", "gaussian5", "gaussian9", "gaussian", "point", "point_inplace", "sharpen", "conv2d"][rng.randint(8)]
        tok = name
        if kind == "passthrough":
            decl.append("%s: passthrough {}" % name)
        elif kind in ("gaussian5", "gaussian9"):
            decl.append("%s: %s { sigma: %.2f }" % (name, kind, rng.uniform(0.4, 3.0)))
        elif kind == "gaussian":
            r = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 11, 15], p=[.1, .15, .15, .15, .15, .1, .08, .05, .04, .03]))
            decl.append("%s: gaussian { sigma: %.2f, radius: %d }" % (name, rng.uniform(0.5, 5.0), r))
        elif kind in ("point", "point_inplace"):
            decl.append("%s: %s { slope: %.2f, offset: %.3f, saturation: %.2f }" % (name, point[rng.randint(3)], rng.uniform(0.5, 1.5), rng.uniform(-0.1, 0.1), rng.uniform(0.0, 2.0)))
            if kind == "point_inplace":
                tok = name + ":image"
        elif kind == "sharpen":
            decl.append("%s: sharpen { amount: %.2f }" % (name, rng.uniform(0.0, 1.5)))
        else:
            k = int(rng.choice([3, 5, 7, 9, 13, 31], p=[.3, .25, .2, .15, .07, .03]))
            decl.append("%s: conv2d { ksize: %d, sigma: %.2f }" % (name, k, rng.uniform(0.6, 4.0)))
        edges.append([src, tok])
        readers[src] = readers.get(src, 0) + 1
        outputs.append(name)
    # every node must reach the output: join the dangling ones into the last node's result
    last = outputs[-1]
    dangling = [o for o in outputs[1:-1] if readers.get(o, 0) == 0]
    j = 0
    for d in dangling:
        name = "j%02d" % j
        j += 1
        decl.append("%s: combination { mix: %.2f }" % (name, rng.uniform(0, 1)))
        edges.append([last, name + ":input_image0"])
        edges.append([d, name + ":input_image1"])
        last = name
    edges.append([last, "output"])
    return "\n".join([" -> ".join(e) for e in edges] + decl)
