"""Known-answer checks that hold for ANY correct implementation of the path.  Each takes
`run(text, img, weights=None) -> out` so the same checks pin the CPU oracle (not gpu)
and the HIP path through the C ABI (gpu).

What is derivable by hand:
  * passthrough is the identity on texel values (shaders/passthrough.comp:7-13),
    for any W x H including sizes that are not multiples of the 16x16 dispatch tile;
  * a filter's impulse response is its kernel, each tap rounded once;
  * degenerate parameters collapse a filter to the identity.
"""
import numpy as np

from oracle import pixel  # only for synthetic inputs and host-side weights

F32, U8 = pixel.FMT_RGBA32F, pixel.FMT_RGBA8

PASSTHROUGH_SIZES = [(1, 1), (17, 13), (64, 64), (65, 3), (3, 65), (512, 512), (250, 131)]


def check_passthrough_identity(run, fmt, W, H):
    x = pixel.fill_synthetic(W, H, fmt, 0x5EED0001)
    out = run("input -> passthrough -> output", x)
    assert out.tobytes() == x.tobytes()


def check_passthrough_preserves_special_floats(run):
    x = np.zeros((4, 8, 4), np.float32)
    x[0, 0] = [np.nan, np.inf, -np.inf, -0.0]
    x[1, 1] = [1e-45, -1e-45, 3.4e38, -3.4e38]      # subnormals must not be flushed
    out = run("input -> passthrough -> output", x)
    assert out.tobytes() == x.tobytes()


def _impulse(W, H, fmt, cx, cy):
    x = np.zeros((H, W, 4), pixel.dtype_of(fmt))
    x[cy, cx, :] = 255 if fmt == U8 else 1.0
    return x


def check_gaussian_impulse(run, golden):
    """Response of gaussian5 (sigma 1) to a unit impulse = outer product of the weights,
    one rounding per pass: out[cy+j, cx+i] = f32(w[|j|] * f32(w[|i|] * 1))."""
    w = golden["gauss_w_s1_r2"]
    W, H, cx, cy = 31, 19, 15, 9
    out = run("input -> gaussian5 -> output\ngaussian5: gaussian5 { sigma: 1.0 }", _impulse(W, H, F32, cx, cy))
    want = np.zeros((H, W, 4), np.float32)
    for j in range(-2, 3):
        for i in range(-2, 3):
            want[cy + j, cx + i, :] = np.float32(w[abs(j)]) * np.float32(w[abs(i)])
    assert out.tobytes() == want.tobytes()


def check_gaussian9_weights(run, golden):
    w = golden["gauss_w_s2_r4"]
    W, H, cx, cy = 40, 21, 20, 10
    out = run("input -> gaussian9 -> output\ngaussian9: gaussian9 { sigma: 2.0 }", _impulse(W, H, F32, cx, cy))
    row = out[cy, cx - 4:cx + 5, 0]
    want = (np.float32(w[0]) * np.array([w[abs(i)] for i in range(-4, 5)], np.float32)).astype(np.float32)
    assert row.tobytes() == want.tobytes()


def check_gaussian_delta_is_identity(run, fmt):
    """sigma <= 0 (also: parameter absent, zero-filled like an unset UBO member,
    render.rs:200-203) is the delta kernel: fma(1, x, 0) = x and fma(0, x, acc) = acc."""
    x = pixel.fill_synthetic(70, 33, fmt, 0x5EED0003)
    out = run("input -> gaussian9 -> output", x)
    assert out.tobytes() == x.tobytes()


def check_sharpen_impulse(run):
    """3x3 cross [0,-a,0; -a,1+4a,-a; 0,-a,0] with a = 0.5."""
    W, H, cx, cy = 21, 17, 10, 8
    out = run("input -> sharpen -> output\nsharpen: sharpen { amount: 0.5 }", _impulse(W, H, F32, cx, cy))
    want = np.zeros((H, W, 4), np.float32)
    want[cy, cx, :] = 3.0
    for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        want[cy + dy, cx + dx, :] = -0.5
    np.testing.assert_array_equal(out, want)        # -0.0 == 0.0 allowed off the cross


def check_sharpen_zero_is_identity(run, fmt):
    x = pixel.fill_synthetic(66, 20, fmt, 0x5EED0002)
    out = run("input -> sharpen -> output", x)          # amount absent -> 0
    np.testing.assert_array_equal(out, x)


def check_conv_impulse_is_flipped_kernel(run):
    """Correlation: out[y,x] = sum w[dy,dx] in[y+dy,x+dx]  =>  an impulse at (cx,cy)
    puts w[dy,dx] at (cx-dx, cy-dy)."""
    K, r = 5, 2
    rng = np.random.RandomState(7)
    w = rng.uniform(-1, 1, (K, K)).astype(np.float32)
    W, H, cx, cy = 23, 19, 11, 9
    out = run("input -> conv2d -> output\nconv2d: conv2d { ksize: 5 }", _impulse(W, H, F32, cx, cy), {"conv2d": w})
    want = np.zeros((H, W, 4), np.float32)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            want[cy - dy, cx - dx, :] = w[dy + r, dx + r]
    np.testing.assert_array_equal(out, want)


def check_grade_saturation_zero_is_grey(run):
    x = pixel.fill_synthetic(64, 16, F32, 0x5EED0002)
    out = run("input -> grade -> output\ngrade: colour_grade { slope: 1.0, offset: 0.0, saturation: 0.0 }", x)
    assert (out[..., 0] == out[..., 1]).all() and (out[..., 1] == out[..., 2]).all()
    assert out[..., 3].tobytes() == x[..., 3].tobytes()          # alpha untouched
    ld = np.longdouble                                  # 64-bit mantissa: product and sum exact, one rounding
    luma = np.float32(0.2126) * x[..., 0]
    luma = (ld(np.float32(0.7152)) * x[..., 1].astype(ld) + luma.astype(ld)).astype(np.float32)
    luma = (ld(np.float32(0.0722)) * x[..., 2].astype(ld) + luma.astype(ld)).astype(np.float32)
    assert np.minimum(np.maximum(luma, 0), 1).astype(np.float32).tobytes() == out[..., 0].tobytes()


def check_unorm8_store_rounds_to_even(run):
    """rgba8 grade with slope 0: every channel stores clamp(offset) -> round-half-even of v*255."""
    for offset, code in ((0.5, 128), (0.1, 26), (1.5, 255), (0.0, 0)):
        x = pixel.fill_synthetic(16, 4, U8, 1)
        text = "input -> grade -> output\ngrade: colour_grade { slope: 0.0, offset: %.4f, saturation: 1.0 }" % offset
        out = run(text, x)
        want = int(np.rint(np.float32(min(max(np.float32(offset), 0), 1)) * np.float32(255)))
        assert want == code
        assert (out[..., :3] == code).all(), (offset, out[0, 0])
        assert (out[..., 3] == x[..., 3]).all()
