// edge_detect.stage.hip -- Sobel gradient magnitude: the `edge_detect` node of the diamond the reference's planner comments
// use as their example (src/vulkan/pipeline_graph.rs:462-468).  A USER stage file: found as {shader_path}/edge_detect.stage.hip
// when a config names the type `edge_detect` (the counterpart of {shader_path}/edge_detect.comp, src/config/config.rs:59-75).
//
// out.c = clamp01(scale * (|gx| + |gy|)) per colour channel, alpha copied;  gx = (NE + 2 E + SE) - (NW + 2 W + SW),
// gy = (SW + 2 S + SE) - (NW + 2 N + NE), every multiply-add a single-rounding fmaf in the order written (the numpy
// restatement in tests/test_gpu_user_stage.py follows the same order; results are bit-identical).
struct Params { float scale; };
static constexpr int RADIUS = 1;

RF_STAGE float sobel1(float nw, float n, float ne, float w, float e, float sw, float s, float se, float scale)
{
    const float right = fmaf(2.0f, e, ne) + se, left = fmaf(2.0f, w, nw) + sw;
    const float below = fmaf(2.0f, s, sw) + se, above = fmaf(2.0f, n, nw) + ne;
    const float g = fabsf(right - left) + fabsf(below - above);
    return fminf(fmaxf(scale * g, 0.0f), 1.0f);
}

RF_STAGE f4 apply(const Params& p, const f4 (&n)[3][3])
{
    return make_float4(sobel1(n[0][0].x, n[0][1].x, n[0][2].x, n[1][0].x, n[1][2].x, n[2][0].x, n[2][1].x, n[2][2].x, p.scale),
                       sobel1(n[0][0].y, n[0][1].y, n[0][2].y, n[1][0].y, n[1][2].y, n[2][0].y, n[2][1].y, n[2][2].y, p.scale),
                       sobel1(n[0][0].z, n[0][1].z, n[0][2].z, n[1][0].z, n[1][2].z, n[2][0].z, n[2][1].z, n[2][2].z, p.scale),
                       n[1][1].w);
}
