// apply_curve.stage.hip -- a user node that READS a storage buffer: RF_BUFFER_IN(block type name, floats) adds a `const float*`
// argument to apply().  The block is found by its TYPE name, ToneCurve (src/vulkan/shader.rs:144-147); shaders/tone_curve.stage.hip
// is a node that fills one.  Each colour channel is mapped through the 256-entry curve with linear interpolation and blended
// with the original by `strength`; alpha is copied.
//   x = clamp(c, 0, 1) * 255;  i = min(int(x), 254);  f = x - i;  m = curve[i] + f * (curve[i+1] - curve[i]);  out = c + strength * (m - c)
struct Params { float strength; };
static constexpr int RADIUS = 0;
RF_BUFFER_IN(ToneCurve, 256);

RF_STAGE float through(float c, float strength, const float* curve)
{
    const float x = fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f;
    int i = (int)x;
    i = i > 254 ? 254 : i;
    const float f = x - (float)i;
    const float a = curve[i], b = curve[i + 1];
    const float m = fmaf(f, b - a, a);
    return fmaf(strength, m - c, c);
}

RF_STAGE void apply(const Params& p, const f4 (&in)[1], f4 (&out)[1], const float* curve)
{
    out[0] = make_float4(through(in[0].x, p.strength, curve), through(in[0].y, p.strength, curve), through(in[0].z, p.strength, curve), in[0].w);
}
