// tone_curve.stage.hip -- a user node that FILLS a storage buffer: RF_BUFFER_OUT(block type name, floats).  The reference finds
// a shader's storage blocks by their TYPE name (src/vulkan/shader.rs:144-147) and a config wires them like images:
//
//     input -> tc -> ac -> output          tc: tone_curve  { gamma: 0.6, lift: 0.05 }
//     tc:ToneCurve -> ac:ToneCurve         ac: apply_curve { strength: 0.8 }          (shaders/apply_curve.stage.hip reads it)
//
// Element i of the buffer is fill(params, i), evaluated for i = 0 .. 255 by a small kernel in front of the node's own, every
// frame.  The image passes through unchanged (like conv2d_weights, the built-in node that writes a ConvWeights block).
//   t = i * (1/255);  c = t + gamma * (t*t - t);  curve[i] = c + lift * (1 - c)      every multiply-add one fmaf, as written.
struct Params { float gamma; float lift; };
static constexpr int RADIUS = 0;
RF_BUFFER_OUT(ToneCurve, 256);

RF_STAGE float fill(const Params& p, int i)
{
    const float t = (float)i * 0.003921569f;
    const float c = fmaf(p.gamma, t * t - t, t);
    return fmaf(p.lift, 1.0f - c, c);
}

RF_STAGE void apply(const Params& p, const f4 (&in)[1], f4 (&out)[1]) { out[0] = in[0]; }
