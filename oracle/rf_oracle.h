/*
 * rf_oracle.h -- CPU restatement of reforge's render-graph hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (librfhip.so)
 * never links, loads or calls it.
 *
 * PARITY UNPINNED (see DESIGN.md "Oracle"): the reference holds no tests,
 * golden vectors or fixtures, cannot be built here (Rust +
 * Vulkan + shaderc, none present) and ships exactly one kernel,
 * shaders/passthrough.comp.  What is pinned by construction:
 *   - passthrough == identity on texel values (shaders/passthrough.comp:7-13)
 *   - UNORM8 load/store conversion rules of Vulkan storage images
 *   - the sRGB transfer functions (IEC 61966-2-1) at the upload/download
 *     blits (src/render.rs:264-313, :406-433)
 * The arithmetic of gaussian / colour_grade / sharpen / conv2d is AUTHORED by
 * this build (the reference names them only in comments:
 * src/config/config_grammar.lalrpop:17, src/vulkan/pipeline_graph.rs:462-468)
 * and is specified in DESIGN.md "Node specifications".
 * Pinned since round 4, against artefacts that are not this build's: the config
 * grammar (tests/golden/grammar_fixtures.json.gz, derived from the reference's
 * config_grammar.lalrpop) and the READING OF GLSL by which the shaders/ .comp files restate
 * this file -- Mesa's GLSL 4.50 compiler + llvmpipe give the same bits as their
 * translation (tests/test_glsl_mesa.py), up to the one rounding of fma() and the
 * exact c / 255 of an rgba8 load, both specified here; Mesa's UNORM8 store rounds
 * ties to even, as rfo_store does.
 *
 * Conventions: images are row-major interleaved RGBA, `pitch` in BYTES.
 * One "invocation" per pixel, one full-frame pass per node
 * (src/vulkan/command.rs:166-242): the loops below deliberately mirror that
 * and do not fuse, tile or vectorise.
 */
#ifndef RF_ORACLE_H
#define RF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFO_FMT_RGBA8   0   /* VK_FORMAT_R8G8B8A8_UNORM      src/main.rs:37 */
#define RFO_FMT_RGBA32F 1   /* VK_FORMAT_R32G32B32A32_SFLOAT src/main.rs:38 */

#define RFO_MAX_RADIUS 15   /* conv2d up to 31x31 */

/* bytes per pixel of a format (4 or 16); 0 for an unknown format */
size_t rfo_bpp(int fmt);

/* number of OpenMP threads used by the node passes (1 = scalar port) */
void rfo_set_threads(int n);
int  rfo_get_threads(void);

/* ---- synthetic inputs -------------------------------- */
uint32_t rfo_hash32(uint32_t seed, uint32_t idx, uint32_t c);
/* fills rows [0,H) of a W-wide image whose first row is global row y0 of a
 * frame of width W: pixel index = (y0+y)*W + x */
void rfo_fill_synthetic(void* img, size_t pitch, int W, int H, int fmt,
                        uint32_t seed, int y0);
/* horizontal + vertical ramps with an impulse at (W/2, Hfull/2) */
void rfo_fill_structured(void* img, size_t pitch, int W, int H, int fmt,
                         int y0, int Hfull);

/* ---- host-side parameter derivation ----------------------------------------- */
/* w[0..radius]: normalised half-kernel, double math rounded once to f32 */
void rfo_gaussian_weights(float sigma, int radius, float* w);
/* centre and side weights of the 3x3 sharpen cross */
float rfo_pulse_slope(float amount, float t);
void rfo_sharpen_weights(float amount, float* centre, float* side);
/* sRGB tables: eotf[c] = linear value of code c; thr[q] = linear value at
 * which the encoded code steps from q to q+1 (q = 0..254) */
void rfo_srgb_tables(float eotf[256], float thr[255]);

/* ---- node passes: in != out unless stated ---------------------------------- */
/* shaders/passthrough.comp:7-13 */
void rfo_passthrough(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                     int W, int H, int fmt);
/* separable gaussian, H pass then V pass on f32 intermediates */
void rfo_gaussian(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                  int W, int H, int fmt, int radius, const float* w);
/* point op; in == out allowed (in-place point-op, pipeline_graph.rs:400-411) */
void rfo_colour_grade(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                      int W, int H, int fmt, float slope, float offset, float saturation);
void rfo_sharpen(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                 int W, int H, int fmt, float amount);
/* dense KxK correlation, K odd <= 31, weights row-major [K][K] */
void rfo_conv2d(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                int W, int H, int fmt, int K, const float* weights);

/* two-input blend out = a + mix*(b-a) (the "combination" node named in
 * pipeline_graph.rs:462-468); out may alias a or b */
void rfo_mix(const void* a, size_t a_pitch, const void* b, size_t b_pitch,
             void* out, size_t out_pitch, int W, int H, int fmt, float mix);
void rfo_split_luma(const void* in, size_t in_pitch, void* luma, size_t luma_pitch, void* chroma, size_t chroma_pitch,
                    int W, int H, int fmt);

/* ---- sRGB boundary (src/render.rs:264-313, :406-433) ------------------------ */
void rfo_upload_srgb8(const uint8_t* rgba, size_t stride, void* img, size_t pitch,
                      int W, int H, int fmt);
void rfo_download_srgb8(const void* img, size_t pitch, uint8_t* rgba, size_t stride,
                        int W, int H, int fmt);

#ifdef __cplusplus
}
#endif
#endif
