// local_contrast.stage.hip -- a user stage of RADIUS 2: the counterpart of a shader that calls imageLoad at neighbouring
// coordinates (shaders/passthrough.comp:9 loads at its own).  From RADIUS 2 on (up to 15) a stage file is a node with a kernel
// of its own and reads its input through a WINDOW: in[0].at(dx, dy), |dx|, |dy| <= RADIUS, clamp-to-edge.
//
//     input -> lc -> output          lc: local_contrast { amount: 0.8 }
//
// out.c = c + amount * (c - mean), mean = the 5x5 box mean around the texel accumulated row by row, left to right, as
// acc = fmaf(0.04, texel, acc) starting from 0 (the restatement in tests/test_gpu_user_node.py follows that order); alpha copied.
struct Params { float amount; };
static constexpr int RADIUS = 2;

RF_STAGE void apply(const Params& p, const Window (&in)[1], f4 (&out)[1])
{
    f4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            const f4 t = in[0].at(dx, dy);
            acc = make_float4(fmaf(0.04f, t.x, acc.x), fmaf(0.04f, t.y, acc.y), fmaf(0.04f, t.z, acc.z), 0.0f);
        }
    const f4 c = in[0].at(0, 0);
    out[0] = make_float4(fmaf(p.amount, c.x - acc.x, c.x), fmaf(p.amount, c.y - acc.y, c.y), fmaf(p.amount, c.z - acc.z, c.z), c.w);
}
