// invert.stage.hip -- the smallest user stage: a point op.  out.rgb = 1 - in.rgb (when `enabled`), alpha copied.
// {shader_path}/invert.stage.hip is found when a config names the type `invert` (src/config/config.rs:59-75).
struct Params { bool enabled; float strength; };
static constexpr int RADIUS = 0;

RF_STAGE f4 apply(const Params& p, f4 c)
{
    if (!p.enabled) return c;
    // c + strength * ((1 - c) - c): strength 1 = the plain negative
    return make_float4(fmaf(p.strength, (1.0f - c.x) - c.x, c.x), fmaf(p.strength, (1.0f - c.y) - c.y, c.y), fmaf(p.strength, (1.0f - c.z) - c.z, c.z), c.w);
}
