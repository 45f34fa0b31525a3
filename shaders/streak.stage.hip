// streak.stage.hip -- a user stage of RADIUS 7: a four-pointed star glow.  From RADIUS 2 on a stage file is a node with a kernel
// of its own and reads its input through a WINDOW, in[0].at(dx, dy), |dx|, |dy| <= RADIUS, clamp-to-edge -- the counterpart of a
// shader that calls imageLoad at neighbouring coordinates (shaders/passthrough.comp:9 loads at its own).  The kernel stages the
// (64 + 14) x (16 + 14) texels a workgroup's 64 x 16 outputs can reach in LDS once; a tap is then one LDS read at a constant offset.
//
//     input -> st -> output          st: streak { amount: 0.6 }
//
// star = the mean of the 28 texels at distances 1..7 to the left, to the right, above and below, accumulated distance by distance in
// that order as acc = fmaf(1/28, texel, acc) from 0 (the restatement in tests/test_gpu_user_node.py follows that order);
// out.c = c + amount * (star - c); alpha copied.
struct Params { float amount; };
static constexpr int RADIUS = 7;

RF_STAGE void apply(const Params& p, const Window (&in)[1], f4 (&out)[1])
{
    const float k = 1.0f / 28.0f;
    f4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int d = 1; d <= 7; ++d) {
        const f4 t[4] = {in[0].at(-d, 0), in[0].at(d, 0), in[0].at(0, -d), in[0].at(0, d)};
        for (int i = 0; i < 4; ++i) acc = make_float4(fmaf(k, t[i].x, acc.x), fmaf(k, t[i].y, acc.y), fmaf(k, t[i].z, acc.z), 0.0f);
    }
    const f4 c = in[0].at(0, 0);
    out[0] = make_float4(fmaf(p.amount, acc.x - c.x, c.x), fmaf(p.amount, acc.y - c.y, c.y), fmaf(p.amount, acc.z - c.z, c.z), c.w);
}
