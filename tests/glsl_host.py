"""Test infrastructure: run a translated GLSL compute shader ON THE HOST.

rf_glsl_translate (the library's translator) gives HIP device source that is plain C++ apart from its attributes; the first half
of reforge_amd/csrc/rf_glsl_dev.h (the GLSL types and built-ins) has no HIP in it either.  This module compiles both with clang++
for x86-64 together with a driver that runs main() once per invocation of the reference's dispatch -- ceil(W/16) x ceil(H/16)
workgroups of the file's local_size (src/vulkan/command.rs:167-168) -- and with the OWN texel conversions of the specification
(rgba8 load c / 255.0f correctly rounded, store clamp * 255 rounded to nearest even, NaN -> 0), so that the CPU suite
can hold translator + prelude to the oracle without a GPU.  What it cannot run: shaders that use workgroup-shared memory or barrier().
Used by tests/ only."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

import reforge_amd as rf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CACHE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "reforge_amd_glsl_host")

DRIVER = r"""
#include <stdint.h>
#include <math.h>
#include <string.h>
#include "rf_glsl_dev.h"
namespace host {
struct T4 { float x, y, z, w; };
struct PxF32 {
    typedef T4 Raw;
    static constexpr int BPP = 16;
    static Raw load(const char* row, unsigned xoff) { Raw r; memcpy(&r, row + xoff, 16); return r; }
    static T4 decode(Raw r) { return r; }
    static T4 texel(float x, float y, float z, float w) { return T4{x, y, z, w}; }
    static void store(char* row, unsigned xoff, T4 v) { memcpy(row + xoff, &v, 16); }
};
static float code(float v) { v = v != v ? 0.0f : v; v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return rintf(v * 255.0f); }
struct PxU8 {
    typedef uint32_t Raw;
    static constexpr int BPP = 4;
    static Raw load(const char* row, unsigned xoff) { Raw r; memcpy(&r, row + xoff, 4); return r; }
#ifndef RFG_SPLIT_FMA
    static T4 decode(Raw r) { return T4{(float)(r & 255u) / 255.0f, (float)((r >> 8) & 255u) / 255.0f, (float)((r >> 16) & 255u) / 255.0f, (float)(r >> 24) / 255.0f}; }
#else   // as llvmpipe decodes UNORM8 (measured, tests/test_glsl_mesa.py): a multiplication by fl(1 / 255) -- the exact quotient for 130 of the 256 codes
    static T4 decode(Raw r) { const float k = 1.0f / 255.0f; return T4{(float)(r & 255u) * k, (float)((r >> 8) & 255u) * k, (float)((r >> 16) & 255u) * k, (float)(r >> 24) * k}; }
#endif
    static T4 texel(float x, float y, float z, float w) { return T4{x, y, z, w}; }
    static void store(char* row, unsigned xoff, T4 v)
    {
        const uint32_t r = (uint32_t)code(v.x) | ((uint32_t)code(v.y) << 8) | ((uint32_t)code(v.z) << 16) | ((uint32_t)code(v.w) << 24);
        memcpy(row + xoff, &r, 4);
    }
};
}
@SOURCE@
using namespace rfglsl;
template <class Px> static void run_all(const GlslFrame& f, const GlslImage* img, void* const* buf, const unsigned char* ubo)
{
    typedef @NS@::RfgInfo I;
    for (unsigned gy = 0; gy < (unsigned)(f.groups_y * I::LY); ++gy)
        for (unsigned gx = 0; gx < (unsigned)(f.groups_x * I::LX); ++gx) {
            if ((int)gy < f.y0 || (int)gy >= f.y1) continue;
            @NS@::RfgShader<Px> s;
            s.gl_WorkGroupID = uvec3{gx / I::LX, gy / I::LY, 0u};
            s.gl_LocalInvocationID = uvec3{gx % I::LX, gy % I::LY, 0u};
            s.gl_NumWorkGroups = uvec3{(unsigned)f.groups_x, (unsigned)f.groups_y, 1u};
            s.gl_GlobalInvocationID = uvec3{gx, gy, 0u};
            s.gl_LocalInvocationIndex = (gy % I::LY) * I::LX + gx % I::LX;
            s.rfg_bind(f, img, buf, ubo);
            s.main();
        }
}
extern "C" int glsl_info(int* v) { typedef @NS@::RfgInfo I; v[0] = I::LX; v[1] = I::LY; v[2] = I::LZ; v[3] = I::NIMG; v[4] = I::NBUF; v[5] = I::UBO; return I::GROUPED ? 1 : 0; }
extern "C" void glsl_run(int u8, const GlslFrame* f, const GlslImage* img, void* const* buf, const unsigned char* ubo)
{
    if (u8) run_all<host::PxU8>(*f, img, buf, ubo);
    else run_all<host::PxF32>(*f, img, buf, ubo);
}
"""


class Frame(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("W", "H", "row_lo", "row_hi", "y0", "y1", "groups_x", "groups_y")] + [("zero", C.c_void_p), ("pad0", C.c_int), ("pad1", C.c_int)]


class Image(C.Structure):
    _fields_ = [("base", C.c_void_p), ("pitch", C.c_ulonglong)]


# as rf_jit.cpp compiles the same source for the device: no contraction, no bit-reinterpreting conversions between float and integer vectors
FLAGS = ["-std=c++17", "-O1", "-ffp-contract=off", "-flax-vector-conversions=integer", "-fwrapv", "-fPIC", "-shared", "-w"]


class HostShader:
    """`text` compiled for the host.  run(images, params, buffers) executes one dispatch."""

    def __init__(self, type_name, text, split_fma=False):
        """split_fma: the two choices Mesa's llvmpipe makes differently from this library's specification -- fma() evaluated with two roundings
        (a * b + c) and UNORM8 texels decoded by a multiplication with fl(1 / 255) -- for tests/test_glsl_mesa.py only"""
        flags = FLAGS + (["-DRFG_SPLIT_FMA"] if split_fma else [])
        self.type_name = type_name
        self.reflection = rf.glsl_reflect(type_name, text)
        src = rf.glsl_translate(type_name, text)
        ns = "rfglsl::" + src.split("\n", 1)[0][3:].strip()
        if self.reflection["grouped"]:
            raise ValueError("%s.comp uses workgroup-shared memory or barrier(): not runnable on the host" % type_name)
        code = DRIVER.replace("@SOURCE@", src).replace("@NS@", ns)
        key = hashlib.sha256((" ".join(flags) + code + open(os.path.join(ROOT, "reforge_amd", "csrc", "rf_glsl_dev.h")).read()).encode()).hexdigest()[:20]
        os.makedirs(CACHE, exist_ok=True)
        so = os.path.join(CACHE, "g_%s.so" % key)
        if not os.path.exists(so):
            cpp = os.path.join(CACHE, "g_%s.cpp" % key)
            with open(cpp, "w") as f:
                f.write(code)
            tmp = so + ".tmp%d" % os.getpid()
            r = subprocess.run([CLANG] + flags + ["-I", os.path.join(ROOT, "reforge_amd", "csrc"), cpp, "-o", tmp, "-lm"],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("host compile of %s.comp failed:\n%s" % (type_name, r.stderr[-3000:]))
            os.replace(tmp, so)
        self.lib = C.CDLL(so)
        info = (C.c_int * 6)()
        self.lib.glsl_info(info)
        self.local_size, self.n_img, self.n_buf, self.ubo_bytes = tuple(info[:3]), info[3], info[4], info[5]
        self._zero = np.zeros(16, np.uint8)

    def run(self, images, params=None, buffers=None, rows=None):
        """images: {variable name: (H, W, 4) array, float32 or uint8, modified in place if the shader writes it};
        params: {uniform member name: value}; buffers: {block type name: numpy array (raw bytes of the block)};
        rows: (y0, y1) frame rows whose invocations run (default: all)."""
        arrs = list(images.values())
        H, W, _ = arrs[0].shape
        u8 = arrs[0].dtype == np.uint8
        fr = Frame(W, H, 0, H - 1, 0 if rows is None else rows[0], H if rows is None else rows[1], (W + 15) // 16, (H + 15) // 16, self._zero.ctypes.data, 0, 0)
        if fr.y1 == H:
            fr.y1 = max(H, ((H + 15) // 16) * self.local_size[1])      # the rows below the frame that the dispatch covers run too (rf_graph.cpp does the same)
        imgs = (Image * max(self.n_img, 1))()
        for i, im in enumerate(self.reflection["images"]):
            a = images.get(im["name"])
            if a is not None:
                assert a.flags["C_CONTIGUOUS"] and a.shape == arrs[0].shape and a.dtype == arrs[0].dtype
                imgs[i] = Image(a.ctypes.data, a.strides[0])
        bufs = (C.c_void_p * max(self.n_buf, 1))()
        keep = []
        for i, b in enumerate(self.reflection["storage_blocks"]):
            a = (buffers or {}).get(b["type_name"])
            if a is None:
                a = np.zeros(b["bytes"], np.uint8)
            assert a.nbytes >= b["bytes"], (b["type_name"], a.nbytes, b["bytes"])
            keep.append(a)
            bufs[i] = a.ctypes.data
        ubo = np.zeros(max((self.ubo_bytes + 7) // 8 * 8, 8), np.uint8)
        for blk in self.reflection["uniform_blocks"]:
            for m in blk["members"]:
                if m["name"] in (params or {}) and m["comps"] == 1 and m["cols"] == 1 and not m["dims"]:
                    v = params[m["name"]]
                    at = blk["base"] + m["offset"]
                    ubo[at:at + 4] = np.frombuffer((np.float32(v) if m["base"] == "f" else np.int32(int(v))).tobytes(), np.uint8)
        self.lib.glsl_run(1 if u8 else 0, C.byref(fr), imgs, bufs, ubo.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return images
