// unsharp_mask.stage.hip -- a user NODE: a stage file that declares its images (RF_INPUTS / RF_OUTPUTS), the counterpart of
// a .comp file with several `uniform image2D` variables, each found by its name (src/vulkan/shader.rs:151-153).  Two input
// images, two output images; a config wires them with descriptor suffixes:
//
//     input -> blur -> um:blurred_image        blur: gaussian9 { sigma: 2.0 }
//     input -> um:input_image                  um: unsharp_mask { amount: 1.5, threshold: 0.02 }
//     um -> output                             (um:mask_image may feed another node, or stay unwired: it is then not stored)
//
// output_image = input + amount * d   where d = input - blurred and |d| >= threshold (per colour channel), else input;
// mask_image   = (|d.r|, |d.g|, |d.b|, 1) where the channel was sharpened, 0 where it was left alone.  Alpha is copied.
// Every multiply-add is one fmaf in the order written (the restatement in tests/test_gpu_user_node.py follows it).
struct Params { float amount; float threshold; };
static constexpr int RADIUS = 0;
RF_INPUTS(input_image, blurred_image);
RF_OUTPUTS(output_image, mask_image);

RF_STAGE void channel(float c, float b, const Params& p, float& o, float& m)
{
    const float d = c - b;
    const bool on = fabsf(d) >= p.threshold;
    o = on ? fmaf(p.amount, d, c) : c;
    m = on ? fabsf(d) : 0.0f;
}

RF_STAGE void apply(const Params& p, const f4 (&in)[2], f4 (&out)[2])
{
    channel(in[0].x, in[1].x, p, out[0].x, out[1].x);
    channel(in[0].y, in[1].y, p, out[0].y, out[1].y);
    channel(in[0].z, in[1].z, p, out[0].z, out[1].z);
    out[0].w = in[0].w;
    out[1].w = 1.0f;
}
