"""Test infrastructure: a `.comp` file run by an INDEPENDENT GLSL implementation -- Mesa's GLSL 4.50 compiler and its llvmpipe CPU back end,
as the image ships them (libgl1-mesa-dri), driven headless by tests/native/mesa_glsl.c.  The reference compiles its filter files with
shaderc and runs them on a Vulkan device; neither exists here, so this is the one place where the text of shaders/*.comp -- and the
translator's reading of the language -- meets a GLSL compiler that is not ours.  MesaShader mirrors tests/glsl_host.HostShader.

What Mesa can and cannot pin: the LANGUAGE (types, conversions, constructors, swizzles, control flow, integer and bit operations, arrays
and structs as values, atomics, shared memory) and every float operation that is a single IEEE operation; not `precise` sequences that
depend on fma() being one rounding if the back end splits it (measured by the tests, not assumed), nor the UNORM8 conversions of rgba8
images, which GL leaves to the implementation within a tolerance."""
import hashlib
import os
import re
import subprocess
import tempfile

import numpy as np

import reforge_amd as rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "mesa_glsl.c")
CACHE = os.path.join(tempfile.gettempdir(), "reforge_amd_mesa_glsl")
_state = {}


def runner():
    """path of the built runner, or None (no compiler / headers / driver: the tests skip)"""
    if "bin" in _state:
        return _state["bin"]
    _state["bin"] = None
    try:
        text = open(SRC).read()
        os.makedirs(CACHE, exist_ok=True)
        exe = os.path.join(CACHE, "mesa_glsl_" + hashlib.sha256(text.encode()).hexdigest()[:16])
        if not os.path.exists(exe):
            tmp = exe + ".tmp%d" % os.getpid()
            r = subprocess.run(["gcc", "-O1", "-o", tmp, SRC, "-ldl"], capture_output=True, text=True)
            if r.returncode != 0:
                _state["why"] = "gcc: " + r.stderr[-500:]
                return None
            os.replace(tmp, exe)
        # one trivial dispatch: is there a driver that gives a 4.5 context?
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "t.comp"), "w").write("#version 450\nlayout (local_size_x = 1) in;\nlayout (binding = 0, rgba32f) uniform image2D image;\n"
                                                       "void main() { imageStore(image, ivec2(0), vec4(1.0)); }\n")
            open(os.path.join(d, "job"), "w").write("image 0 rgba32f - %s\n" % os.path.join(d, "o.raw"))
            r = subprocess.run([exe, os.path.join(d, "t.comp"), "1", "1", "1", "1", os.path.join(d, "job")], capture_output=True, text=True, timeout=300)
            if r.returncode != 0 or not os.path.exists(os.path.join(d, "o.raw")) or np.fromfile(os.path.join(d, "o.raw"), np.float32).tolist() != [1.0] * 4:
                _state["why"] = "mesa_glsl rc=%d: %s" % (r.returncode, r.stderr[-500:])
                return None
            _state["version"] = r.stderr.strip().split("\n")[0]
        _state["bin"] = exe
    except Exception as e:      # noqa: BLE001 -- whatever is missing, the tests skip with the reason
        _state["why"] = repr(e)
    return _state["bin"]


def why_not():
    return _state.get("why", "")


def version():
    return _state.get("version", "")


class MesaCompileError(Exception):
    pass


class MesaShader:
    """`text` compiled and run by Mesa.  run(images, params, buffers) executes the reference's dispatch: ceil(W/16) x ceil(H/16) workgroups."""

    def __init__(self, type_name, text):
        self.type_name, self.text = type_name, text
        self.reflection = rf.glsl_reflect(type_name, text)

    def _with_format(self, fmt):
        """GL wants an image's format qualifier to be the format of the texture bound to it (the reference binds the graph's format whatever
        the file says: passthrough.comp says rgba8 and runs on rgba32f images) -- so the qualifier of every image declaration is rewritten"""
        def fix(m):
            return re.sub(r"\b(rgba32f|rgba16f|rgba8|rgba16|rgba8_snorm|r32f)\b", fmt, m.group(0))
        return re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(?:readonly\s+|writeonly\s+|coherent\s+|restrict\s+|volatile\s+)*image2D", fix, self.text)

    def run(self, images, params=None, buffers=None, groups=None):
        """images: {variable name: (H, W, 4) float32 or uint8 array}; returns {name: array after the dispatch} for every image the shader may
        write, and updates `buffers` ({block type name: numpy array of its bytes}) in place for every block it may write"""
        arrs = list(images.values())
        H, W, _ = arrs[0].shape
        u8 = arrs[0].dtype == np.uint8
        fmt = "rgba8" if u8 else "rgba32f"
        exe = runner()
        assert exe, why_not()
        with tempfile.TemporaryDirectory() as d:
            comp = os.path.join(d, self.type_name + ".comp")
            open(comp, "w").write(self._with_format(fmt))
            job, outs = [], {}
            for im in self.reflection["images"]:
                a = images.get(im["name"])
                src = "-"
                if a is not None:
                    assert a.shape == arrs[0].shape and a.dtype == arrs[0].dtype
                    src = os.path.join(d, "in_%s.raw" % im["name"])
                    np.ascontiguousarray(a).tofile(src)
                if im.get("sampled"):
                    job.append("sampler %d %s %s" % (im["binding"], fmt, src))
                    continue
                dst = "-"
                if not im["readonly"]:
                    dst = os.path.join(d, "out_%s.raw" % im["name"])
                    outs[im["name"]] = dst
                job.append("image %d %s %s %s" % (im["binding"], fmt, src, dst))
            for blk in self.reflection["uniform_blocks"]:
                ubo = np.zeros(max((blk["bytes"] + 15) // 16 * 16, 16), np.uint8)
                for m in blk["members"]:
                    if m["name"] in (params or {}) and m["comps"] == 1 and m["cols"] == 1 and not m["dims"]:
                        v = params[m["name"]]
                        ubo[m["offset"]:m["offset"] + 4] = np.frombuffer((np.float32(v) if m["base"] == "f" else np.int32(int(v))).tobytes(), np.uint8)
                path = os.path.join(d, "ubo_%d.bin" % blk["binding"])
                ubo.tofile(path)
                job.append("ubo %d %s" % (blk["binding"], path))
            bouts = {}
            for blk in self.reflection["storage_blocks"]:
                a = (buffers or {}).get(blk["type_name"])
                src = "-"
                if a is not None:
                    src = os.path.join(d, "ssbo_%d.bin" % blk["binding"])
                    np.ascontiguousarray(a).view(np.uint8).tofile(src)
                dst = "-"
                if not blk["readonly"] and a is not None:
                    dst = os.path.join(d, "ssbo_%d.out" % blk["binding"])
                    bouts[blk["type_name"]] = dst
                job.append("ssbo %d %d %s %s" % (blk["binding"], max(blk["bytes"], 4), src, dst))
            open(os.path.join(d, "job"), "w").write("\n".join(job) + "\n")
            gx, gy = groups or ((W + 15) // 16, (H + 15) // 16)
            r = subprocess.run([exe, comp, str(W), str(H), str(gx), str(gy), os.path.join(d, "job")], capture_output=True, text=True, timeout=600)
            self.last_log = r.stderr      # (scripts/mesa_baseline.py reads the runner's timing line)
            if r.returncode == 3:
                raise MesaCompileError(r.stderr)
            assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
            result = {name: np.fromfile(path, arrs[0].dtype).reshape(arrs[0].shape) for name, path in outs.items()}
            for name, path in bouts.items():
                b = buffers[name].view(np.uint8).reshape(-1)
                got = np.fromfile(path, np.uint8)
                b[:got.size] = got
            return result


class FileGraph:
    """A whole config run the REFERENCE'S WAY with the filter files as the kernels: the config is parsed, the nodes ordered and the images
    aliased by the oracle's restatement of the reference's planner (oracle/graph.py: config.rs, pipeline_graph.rs:358-497), and every node
    is ONE dispatch of `shaders/{type}.comp` over ceil(W/16) x ceil(H/16) workgroups (command.rs:166-194) -- executed by `how`:
    "mesa" (Mesa's GLSL compiler + llvmpipe) or "host" (rf_glsl.cpp's translation compiled for the host, fma() split like llvmpipe's)."""

    def __init__(self, text, img, how, shader_dir):
        from oracle import graph as ograph
        from oracle import pixel
        from tests.glsl_host import HostShader
        H, W, _ = img.shape
        g = ograph.GraphOracle(text, W, H, pixel.fmt_of(img))
        g.upload_raw(img)
        shaders = {}

        def run_node(info):
            def image(resource):
                return g.images[ograph._remap(resource, g.reuse)]
            if info.type not in shaders:
                with open(os.path.join(shader_dir, info.type + ".comp")) as f:
                    src = f.read()
                shaders[info.type] = MesaShader(info.type, src) if how == "mesa" else HostShader(info.type, src, split_fma=True)
            sh = shaders[info.type]
            name_of = {im["binding"]: im["name"] for im in sh.reflection["images"]}
            block_of = {b["binding"]: b["type_name"] for b in sh.reflection["storage_blocks"]}
            images, written = {}, {}
            for r, b in info.input_images:
                images[name_of[b]] = image(r)
            for r, b in info.output_images:
                images.setdefault(name_of[b], image(r))
                written[name_of[b]] = image(r)
            buffers = {block_of[b]: g.ssbos[ograph._remap(r, g.ssbo_remap)] for r, b in info.input_ssbos + info.output_ssbos}
            out = sh.run(images, dict(info.params), buffers or None)
            if how == "mesa":
                for name, arr in written.items():
                    arr[...] = out[name]

        for layer in g.layers:
            for node in layer:
                run_node(g.infos[node])
        self.result = g.download_raw()


def mesa_layout(text):
    """{(kind, member name): (offset, array stride, matrix stride)} and {(kind, block name): (binding, bytes)} of the blocks of `text`, as Mesa's
    linker lays them out (program interface queries): an independent reading of std140 / std430 for rf_glsl_reflect to be held against"""
    exe = runner()
    assert exe, why_not()
    with tempfile.TemporaryDirectory() as d:
        comp = os.path.join(d, "layout.comp")
        open(comp, "w").write(text)
        open(os.path.join(d, "job"), "w").write("")
        r = subprocess.run([exe, comp, "1", "1", "1", "1", os.path.join(d, "job")], capture_output=True, text=True, timeout=300, env=dict(os.environ, RF_MESA_REFLECT="1"))
        if r.returncode == 3:
            raise MesaCompileError(r.stderr)
        assert r.returncode == 0, r.stderr[-1000:]
    members, blocks = {}, {}
    for line in r.stdout.split("\n"):
        w = line.split()
        if w[:1] == ["block"]:
            blocks[(w[1], w[2])] = (int(w[4]), int(w[6]))
        elif w[:1] == ["member"]:
            members[(w[1], w[2])] = (int(w[4]), int(w[6]), int(w[8]))
    return members, blocks
