"""CPU execution of USER filter types ({shader_path}/{type}.stage.hip) for the graph oracle.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

A stage file is plain C++ (struct Params, RADIUS, apply(), optionally fill(); fmaf / fminf / fabsf ...), so the oracle
compiles THE FILE ITSELF for the host with g++ (-O2 -ffp-contract=off: every multiply-add stays the fmaf the file wrote) and
calls it texel by texel.  This is NOT an independent statement of what a file computes -- tests/test_gpu_user_stage.py and
tests/test_gpu_user_node.py check the shipped files against exact-rational restatements for that.  What it buys is an oracle
for the EXECUTOR around user types: graphs generated at random that mix them with built-in nodes -- fusion with neighbours,
aliasing, in-place writes, several outputs, storage-buffer edges, row strips -- have an expected result again
(tests/util.py random_dag(user=True), scripts/fuzz_graphs.py).

"Reflection" is restated here in Python (the product's is reforge_amd/csrc/rf_user.cpp): members of struct Params with their
offsets, RADIUS, RF_INPUTS / RF_OUTPUTS (bindings: inputs 0.., outputs after them, a name on both sides shares its input's),
RF_BUFFER_IN / RF_BUFFER_OUT (bindings after the images'), RADIUS > 0 in a node = inputs read through `Window::at`.  Ref: src/vulkan/shader.rs:106-160 (reflect_descriptors).
"""
import ctypes as C
import hashlib
import os
import re
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_user")

_HARNESS = r'''
#include <math.h>
#include <stdint.h>
#include <string.h>
struct f4 { float x, y, z, w; };
static inline f4 make_float4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
struct Window {       /* RADIUS > 0 in a node: the neighbourhood of an input image, clamp-to-edge (rf_user_dev.h) */
    const f4* img; int W, H, x, y;
    f4 at(int dx, int dy) const
    {
        int xx = x + dx, yy = y + dy;
        xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        return img[(long)yy * W + xx];
    }
};
#define RF_STAGE static inline
#define RF_INPUTS(...) static_assert(true, "")
#define RF_OUTPUTS(...) static_assert(true, "")
#define RF_BUFFER_IN(...) static_assert(true, "")
#define RF_BUFFER_OUT(...) static_assert(true, "")
#line 1 "%(name)s.stage.hip"
%(text)s
#line 1000 "harness"
static_assert(sizeof(Params) == %(psize)d, "struct Params is not laid out as the oracle computed");
#if FORM == 0      /* point op: apply(p, c) */
extern "C" void rf_run(const void* pp, const float* src, float* dst, int W, int H)
{
    Params p; memcpy(&p, pp, sizeof(p));
    const f4* s = (const f4*)src; f4* d = (f4*)dst;
    for (long i = 0; i < (long)W * H; ++i) d[i] = apply(p, s[i]);
}
#elif FORM == 1    /* 3x3 neighbourhood, clamp-to-edge: apply(p, n[3][3]) */
extern "C" void rf_run(const void* pp, const float* src, float* dst, int W, int H)
{
    Params p; memcpy(&p, pp, sizeof(p));
    const f4* s = (const f4*)src; f4* d = (f4*)dst;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            f4 n[3][3];
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    n[dy + 1][dx + 1] = s[(long)yy * W + xx];
                }
            d[(long)y * W + x] = apply(p, n);
        }
}
#else              /* a node: NI inputs, NO outputs, optionally a buffer read and a buffer filled */
extern "C" void rf_node(const void* pp, const float* const* in, float* const* out, const float* buf, int W, int H)
{
    Params p; memcpy(&p, pp, sizeof(p));
    for (long i = 0; i < (long)W * H; ++i) {
        f4 o[NO];
#if WINDOWS
        Window a[NI];
        for (int k = 0; k < NI; ++k) { Window w = {(const f4*)in[k], W, H, (int)(i %% W), (int)(i / W)}; a[k] = w; }
#else
        f4 a[NI];
        for (int k = 0; k < NI; ++k) a[k] = ((const f4*)in[k])[i];
#endif
        for (int k = 0; k < NO; ++k) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#if HAS_BUF_IN
        apply(p, a, o, buf);
#else
        apply(p, a, o);
#endif
        for (int k = 0; k < NO; ++k) if (out[k]) ((f4*)out[k])[i] = o[k];
    }
}
#if FILL > 0
extern "C" void rf_fill(const void* pp, float* buf)
{
    Params p; memcpy(&p, pp, sizeof(p));
    for (int i = 0; i < FILL; ++i) buf[i] = fill(p, i);
}
#endif
#endif
'''


def _strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


class UserType:
    """what reflection of one stage file yields + the compiled host functions"""

    def __init__(self, name, path):
        self.name, self.path = name, path
        self.text = open(path).read()
        t = _strip_comments(self.text)
        m = re.search(r"RADIUS\s*=\s*(\d+)", t)
        self.radius = int(m.group(1))
        body = re.search(r"struct\s+Params\s*\{(.*?)\}", t, re.S).group(1)
        self.params, off, align = [], 0, 1            # (name, type, offset)
        for decl in body.split(";"):
            parts = decl.split()
            if not parts:
                continue
            ty, nm = parts
            size = 1 if ty == "bool" else 4
            off = (off + size - 1) // size * size
            self.params.append((nm, {"float": "f32", "int": "i32", "bool": "bool"}[ty], off))
            off += size
            align = max(align, size)
        self.params_size = max(1, (off + align - 1) // align * align)

        def names(kw):
            m = re.search(kw + r"\s*\(([^)]*)\)", t)
            return [x.strip() for x in m.group(1).split(",")] if m else None
        ins, outs = names("RF_INPUTS"), names("RF_OUTPUTS")
        self.buf_in = names("RF_BUFFER_IN")            # [type name, count]
        self.buf_out = names("RF_BUFFER_OUT")
        self.multi = any(v is not None for v in (ins, outs, self.buf_in, self.buf_out)) or self.radius >= 2
        self.inputs = ins or ["input_image"]
        self.outputs = outs or ["output_image"]
        self.images, nxt = {}, len(self.inputs)
        for i, nm in enumerate(self.inputs):
            self.images[nm] = i
        self.out_binding = []
        for nm in self.outputs:
            if nm in self.inputs:
                self.out_binding.append(self.inputs.index(nm))     # same name = same binding = in place
            else:
                self.images[nm] = nxt
                self.out_binding.append(nxt)
                nxt += 1
        self.buffers = {}
        for b in (self.buf_in, self.buf_out):
            if b:
                self.buffers[b[0]] = (nxt, int(b[1]) * 4)
                nxt += 1
        self._lib = None

    def node_type(self):
        """the entry oracle/graph.py's NODE_TYPES wants"""
        d = {"images": dict(self.images), "params": {n: ty for n, ty, _ in self.params}, "user": self}
        if self.buffers:
            d["buffers"] = dict(self.buffers)
        return d

    def lib(self):
        if self._lib is None:
            form = 2 if self.multi else (1 if self.radius else 0)
            src = _HARNESS % {"name": self.name, "text": self.text, "psize": self.params_size}
            flags = ["-DFORM=%d" % form, "-DNI=%d" % len(self.inputs), "-DNO=%d" % len(self.outputs), "-DHAS_BUF_IN=%d" % (1 if self.buf_in else 0), "-DWINDOWS=%d" % (1 if self.radius > 0 else 0),
                     "-DFILL=%d" % (int(self.buf_out[1]) if self.buf_out else 0)]
            key = hashlib.sha256((src + " ".join(flags)).encode()).hexdigest()[:20]
            os.makedirs(_BUILD, exist_ok=True)
            so = os.path.join(_BUILD, "%s_%s.so" % (self.name, key))
            if not os.path.exists(so):
                cpp = so[:-3] + ".cpp"
                with open(cpp, "w") as f:
                    f.write(src)
                tmp = so + ".tmp%d" % os.getpid()
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-x", "c++"] + flags + ["-o", tmp, cpp])
                os.replace(tmp, so)
            self._lib = C.CDLL(so)
        return self._lib

    def params_block(self, values):
        blk = bytearray(self.params_size)
        for nm, ty, off in self.params:
            v = values.get(nm, 0)
            if ty == "f32":
                struct.pack_into("<f", blk, off, float(v))
            elif ty == "i32":
                struct.pack_into("<i", blk, off, int(v))
            else:
                blk[off] = 1 if v else 0
        return bytes(blk)


# ---- texel formats in numpy: what imageLoad / imageStore do (shaders/passthrough.comp:9,:12) -------------------
def decode(img):
    """stored image -> float32 (H, W, 4): rgba32f as is; UNORM8 code / 255, correctly rounded (IEEE float32 division)"""
    if img.dtype == np.float32:
        return np.ascontiguousarray(img)
    return (img.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def encode(vals, dst):
    """float32 (H, W, 4) -> the stored image `dst` (in place): rgba32f bit copy; UNORM8 clamp (NaN -> 0), x255 in float32, round to nearest even"""
    if dst.dtype == np.float32:
        dst[...] = vals
        return
    v = np.where(np.isnan(vals), np.float32(0), vals)
    v = np.minimum(np.maximum(v, np.float32(0)), np.float32(1)).astype(np.float32)
    dst[...] = np.rint(v * np.float32(255.0)).astype(np.uint8)


def run(ut, params, srcs, dsts, buf_in=None, buf_out=None):
    """one user node over whole images.  srcs: stored images in input-declaration order; dsts: one per declared output (None =
    not wired).  Every input is decoded BEFORE any output is stored, so a dst that is also a src (in place) is safe."""
    L = ut.lib()
    H, W, _ = srcs[0].shape
    blk = C.create_string_buffer(ut.params_block(params), ut.params_size)
    ins = [decode(s) for s in srcs]
    fp = C.POINTER(C.c_float)
    if not ut.multi:
        out = np.empty((H, W, 4), np.float32)
        L.rf_run(blk, ins[0].ctypes.data_as(fp), out.ctypes.data_as(fp), C.c_int(W), C.c_int(H))
        encode(out, dsts[0])
        return
    if ut.buf_out and buf_out is not None:
        L.rf_fill(blk, buf_out.ctypes.data_as(fp))
    outs = [np.empty((H, W, 4), np.float32) if d is not None else None for d in dsts]
    in_arr = (fp * len(ins))(*[a.ctypes.data_as(fp) for a in ins])
    out_arr = (fp * len(outs))(*[(a.ctypes.data_as(fp) if a is not None else fp()) for a in outs])
    L.rf_node(blk, in_arr, out_arr, buf_in.ctypes.data_as(fp) if buf_in is not None else fp(), C.c_int(W), C.c_int(H))
    for a, d in zip(outs, dsts):
        if d is not None:
            encode(a, d)
