"""ctypes binding of oracle/liboracle.so (rf_oracle.c) on numpy arrays.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Images are numpy arrays of shape (H, W, 4), dtype uint8 (rgba8) or float32
(rgba32f), C-contiguous along the last two axes; the row pitch is taken from
the array's strides so ghost-row views work.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

FMT_RGBA8 = 0
FMT_RGBA32F = 1
MAX_RADIUS = 15


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or (
        os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "rf_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, sz, i32, u32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_float
        fp = C.POINTER(C.c_float)
        L.rfo_bpp.restype = sz
        L.rfo_bpp.argtypes = [i32]
        L.rfo_set_threads.argtypes = [i32]
        L.rfo_get_threads.restype = i32
        L.rfo_hash32.restype = u32
        L.rfo_hash32.argtypes = [u32, u32, u32]
        L.rfo_fill_synthetic.argtypes = [vp, sz, i32, i32, i32, u32, i32]
        L.rfo_fill_structured.argtypes = [vp, sz, i32, i32, i32, i32, i32]
        L.rfo_gaussian_weights.argtypes = [f32, i32, fp]
        L.rfo_sharpen_weights.argtypes = [f32, fp, fp]
        L.rfo_pulse_slope.restype = f32
        L.rfo_pulse_slope.argtypes = [f32, f32]
        L.rfo_srgb_tables.argtypes = [fp, fp]
        L.rfo_passthrough.argtypes = [vp, sz, vp, sz, i32, i32, i32]
        L.rfo_gaussian.argtypes = [vp, sz, vp, sz, i32, i32, i32, i32, fp]
        L.rfo_colour_grade.argtypes = [vp, sz, vp, sz, i32, i32, i32, f32, f32, f32]
        L.rfo_sharpen.argtypes = [vp, sz, vp, sz, i32, i32, i32, f32]
        L.rfo_conv2d.argtypes = [vp, sz, vp, sz, i32, i32, i32, i32, fp]
        L.rfo_mix.argtypes = [vp, sz, vp, sz, vp, sz, i32, i32, i32, f32]
        L.rfo_split_luma.argtypes = [vp, sz, vp, sz, vp, sz, i32, i32, i32]
        L.rfo_upload_srgb8.argtypes = [vp, sz, vp, sz, i32, i32, i32]
        L.rfo_download_srgb8.argtypes = [vp, sz, vp, sz, i32, i32, i32]
        _lib = L
    return _lib


def fmt_of(img):
    if img.dtype == np.uint8:
        return FMT_RGBA8
    if img.dtype == np.float32:
        return FMT_RGBA32F
    raise TypeError("image dtype must be uint8 or float32, got %s" % img.dtype)


def dtype_of(fmt):
    return np.uint8 if fmt == FMT_RGBA8 else np.float32


def _chk(img):
    assert img.ndim == 3 and img.shape[2] == 4, img.shape
    isz = img.dtype.itemsize
    assert img.strides[2] == isz and img.strides[1] == 4 * isz, "pixels must be contiguous"
    assert img.strides[0] >= img.shape[1] * 4 * isz
    return img.ctypes.data, img.strides[0], img.shape[1], img.shape[0], fmt_of(img)


def new_image(W, H, fmt):
    return np.zeros((H, W, 4), dtype=dtype_of(fmt))


def set_threads(n):
    lib().rfo_set_threads(int(n))


def hash32(seed, idx, c):
    return lib().rfo_hash32(seed & 0xFFFFFFFF, idx & 0xFFFFFFFF, c)


def fill_synthetic(W, H, fmt, seed, y0=0):
    img = new_image(W, H, fmt)
    p, pitch, _, _, _ = _chk(img)
    lib().rfo_fill_synthetic(p, pitch, W, H, fmt, seed & 0xFFFFFFFF, y0)
    return img


def fill_structured(W, H, fmt, y0=0, Hfull=None):
    img = new_image(W, H, fmt)
    p, pitch, _, _, _ = _chk(img)
    lib().rfo_fill_structured(p, pitch, W, H, fmt, y0, H if Hfull is None else Hfull)
    return img


def gaussian_weights(sigma, radius):
    w = np.zeros(radius + 1, dtype=np.float32)
    lib().rfo_gaussian_weights(float(sigma), int(radius), w.ctypes.data_as(C.POINTER(C.c_float)))
    return w


def sharpen_weights(amount):
    c, s = C.c_float(), C.c_float()
    lib().rfo_sharpen_weights(float(amount), C.byref(c), C.byref(s))
    return c.value, s.value


def pulse_slope(amount, t):
    return float(lib().rfo_pulse_slope(float(amount), float(t)))


def srgb_tables():
    eotf = np.zeros(256, dtype=np.float32)
    thr = np.zeros(255, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().rfo_srgb_tables(eotf.ctypes.data_as(fp), thr.ctypes.data_as(fp))
    return eotf, thr


def _pair(src, dst):
    ps, pis, W, H, fmt = _chk(src)
    pd, pid, W2, H2, fmt2 = _chk(dst)
    assert (W, H, fmt) == (W2, H2, fmt2)
    return ps, pis, pd, pid, W, H, fmt


def passthrough(src, dst=None):
    dst = np.empty_like(src) if dst is None else dst
    lib().rfo_passthrough(*_pair(src, dst))
    return dst


def gaussian(src, radius, sigma=None, weights=None, dst=None):
    dst = np.empty_like(src) if dst is None else dst
    w = gaussian_weights(sigma, radius) if weights is None else np.ascontiguousarray(weights, np.float32)
    assert w.shape == (radius + 1,)
    ps, pis, pd, pid, W, H, fmt = _pair(src, dst)
    lib().rfo_gaussian(ps, pis, pd, pid, W, H, fmt, radius, w.ctypes.data_as(C.POINTER(C.c_float)))
    return dst


def colour_grade(src, slope, offset, saturation, dst=None):
    dst = np.empty_like(src) if dst is None else dst
    ps, pis, pd, pid, W, H, fmt = _pair(src, dst)
    lib().rfo_colour_grade(ps, pis, pd, pid, W, H, fmt, float(slope), float(offset), float(saturation))
    return dst


def sharpen(src, amount, dst=None):
    dst = np.empty_like(src) if dst is None else dst
    ps, pis, pd, pid, W, H, fmt = _pair(src, dst)
    lib().rfo_sharpen(ps, pis, pd, pid, W, H, fmt, float(amount))
    return dst


def conv2d(src, weights, dst=None):
    dst = np.empty_like(src) if dst is None else dst
    w = np.ascontiguousarray(weights, np.float32)
    K = w.shape[0]
    assert w.shape == (K, K) and K % 2 == 1 and K // 2 <= MAX_RADIUS
    ps, pis, pd, pid, W, H, fmt = _pair(src, dst)
    lib().rfo_conv2d(ps, pis, pd, pid, W, H, fmt, K, w.ctypes.data_as(C.POINTER(C.c_float)))
    return dst


def mix(a, b, t, dst=None):
    dst = np.empty_like(a) if dst is None else dst
    pa, pia, pd, pid, W, H, fmt = _pair(a, dst)
    pb, pib, _, _, _, _, _ = _pair(b, dst)
    lib().rfo_mix(pa, pia, pb, pib, pd, pid, W, H, fmt, float(t))
    return dst


def split_luma(src, luma=None, chroma=None):
    """one input, two outputs (either may be None = not wired); returns (luma, chroma)"""
    ps, pis, W, H, fmt = _chk(src)
    pl = pil = pc = pic = 0
    if luma is not None:
        pl, pil, _, _, _ = _chk(luma)
    if chroma is not None:
        pc, pic, _, _, _ = _chk(chroma)
    lib().rfo_split_luma(ps, pis, pl, pil, pc, pic, W, H, fmt)
    return luma, chroma


def upload_srgb8(rgba, fmt):
    """rgba: (H, W, 4) uint8 sRGB -> linear image of `fmt` (render.rs:264-313)."""
    rgba = np.ascontiguousarray(rgba, np.uint8)
    H, W, _ = rgba.shape
    img = new_image(W, H, fmt)
    p, pitch, _, _, _ = _chk(img)
    lib().rfo_upload_srgb8(rgba.ctypes.data, rgba.strides[0], p, pitch, W, H, fmt)
    return img


def download_srgb8(img):
    """linear image -> (H, W, 4) uint8 sRGB (render.rs:406-433)."""
    p, pitch, W, H, fmt = _chk(img)
    out = np.zeros((H, W, 4), dtype=np.uint8)
    lib().rfo_download_srgb8(p, pitch, out.ctypes.data, out.strides[0], W, H, fmt)
    return out
