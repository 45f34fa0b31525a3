/*
 * rf_oracle.c -- scalar CPU restatement of reforge's render-graph hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED for the authored nodes -- read the
 * header of rf_oracle.h first.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; every fused
 * multiply-add below is an explicit fmaf(), so the result does not depend on
 * the compiler's contraction choices).
 */
#include "rf_oracle.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void rfo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int  rfo_get_threads(void)  { return g_threads; }

size_t rfo_bpp(int fmt)
{
    if (fmt == RFO_FMT_RGBA8)   return 4;
    if (fmt == RFO_FMT_RGBA32F) return 16;
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Texel access: what imageLoad / imageStore do on a storage image            */
/* (shaders/passthrough.comp:9,:12).  UNORM8: load c/255, store               */
/* clamp -> x255 -> round-to-nearest-even; NaN stores 0.  SFLOAT: bit copy.    */
/* ------------------------------------------------------------------------- */

typedef struct { float c[4]; } px4;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline float unorm8_to_f32(uint8_t c) { return (float)c / 255.0f; }

static inline uint8_t f32_to_unorm8(float v)
{
    if (!(v > 0.0f)) return 0;          /* also NaN */
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)rintf(v * 255.0f);  /* default rounding mode: nearest even */
}

static inline px4 load_px(const void* img, size_t pitch, int fmt, int x, int y)
{
    px4 p;
    const uint8_t* row = (const uint8_t*)img + (size_t)y * pitch;
    if (fmt == RFO_FMT_RGBA8) {
        const uint8_t* q = row + (size_t)x * 4;
        for (int k = 0; k < 4; ++k) p.c[k] = unorm8_to_f32(q[k]);
    } else {
        memcpy(p.c, row + (size_t)x * 16, 16);
    }
    return p;
}

static inline void store_px(void* img, size_t pitch, int fmt, int x, int y, px4 p)
{
    uint8_t* row = (uint8_t*)img + (size_t)y * pitch;
    if (fmt == RFO_FMT_RGBA8) {
        uint8_t* q = row + (size_t)x * 4;
        for (int k = 0; k < 4; ++k) q[k] = f32_to_unorm8(p.c[k]);
    } else {
        memcpy(row + (size_t)x * 16, p.c, 16);
    }
}

/* ------------------------------------------------------------------------- */
/* Synthetic inputs                                             */
/* ------------------------------------------------------------------------- */

uint32_t rfo_hash32(uint32_t seed, uint32_t idx, uint32_t c)
{
    uint32_t h = seed ^ ((idx * 4u + c) * 0x9E3779B1u);
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

void rfo_fill_synthetic(void* img, size_t pitch, int W, int H, int fmt,
                        uint32_t seed, int y0)
{
    for (int y = 0; y < H; ++y) {
        uint8_t* row = (uint8_t*)img + (size_t)y * pitch;
        for (int x = 0; x < W; ++x) {
            uint32_t idx = (uint32_t)(y0 + y) * (uint32_t)W + (uint32_t)x;
            for (uint32_t k = 0; k < 4; ++k) {
                uint32_t u = rfo_hash32(seed, idx, k);
                if (fmt == RFO_FMT_RGBA8) {
                    row[(size_t)x * 4 + k] = (uint8_t)(u >> 24);
                } else {
                    float v = (float)(u >> 8) * 0x1p-24f;   /* exact, in [0,1) */
                    memcpy(row + (size_t)x * 16 + k * 4, &v, 4);
                }
            }
        }
    }
}

void rfo_fill_structured(void* img, size_t pitch, int W, int H, int fmt,
                         int y0, int Hfull)
{
    for (int y = 0; y < H; ++y) {
        int gy = y0 + y;
        uint8_t* row = (uint8_t*)img + (size_t)y * pitch;
        for (int x = 0; x < W; ++x) {
            uint8_t code[4];
            code[0] = (uint8_t)(x & 255);
            code[1] = (uint8_t)(gy & 255);
            code[2] = (uint8_t)((x + gy) & 255);
            code[3] = 255;
            if (x == W / 2 && gy == Hfull / 2) code[0] = code[1] = code[2] = 255;
            for (int k = 0; k < 4; ++k) {
                if (fmt == RFO_FMT_RGBA8) {
                    row[(size_t)x * 4 + k] = code[k];
                } else {
                    float v = unorm8_to_f32(code[k]);
                    memcpy(row + (size_t)x * 16 + k * 4, &v, 4);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Host-side parameter derivation                                              */
/* ------------------------------------------------------------------------- */

void rfo_gaussian_weights(float sigma, int radius, float* w)
{
    if (!(sigma > 0.0f)) {              /* sigma <= 0 or NaN: the delta kernel */
        w[0] = 1.0f;
        for (int i = 1; i <= radius; ++i) w[i] = 0.0f;
        return;
    }
    double e[RFO_MAX_RADIUS + 1];
    double s2 = 2.0 * (double)sigma * (double)sigma;
    for (int i = 0; i <= radius; ++i) e[i] = exp(-(double)(i * i) / s2);
    double sum = e[0];
    for (int i = 1; i <= radius; ++i) sum += 2.0 * e[i];
    for (int i = 0; i <= radius; ++i) w[i] = (float)(e[i] / sum);
}

void rfo_sharpen_weights(float amount, float* centre, float* side)
{
    *centre = fmaf(4.0f, amount, 1.0f);
    *side   = -amount;
}

/* `pulse` node: a colour grade whose slope follows the frame time -- slope = fma(amount, frac(t), 1) (one rounding; the
 * fractional part of an f32 is exact), offset 0, saturation 1.  `t` is the member `phase_rf_time`, which the host
 * overwrites with the seconds since start every frame (src/render.rs:212-223). */
float rfo_pulse_slope(float amount, float t)
{
    float f = t - floorf(t);
    if (!(f >= 0.0f && f < 1.0f)) f = 0.0f;     /* NaN / infinite time: no pulse */
    return fmaf(amount, f, 1.0f);
}

static double srgb_eotf_d(double cs)
{
    return cs <= 0.04045 ? cs / 12.92 : pow((cs + 0.055) / 1.055, 2.4);
}

void rfo_srgb_tables(float eotf[256], float thr[255])
{
    for (int c = 0; c < 256; ++c) eotf[c] = (float)srgb_eotf_d((double)c / 255.0);
    for (int q = 0; q < 255; ++q) thr[q]  = (float)srgb_eotf_d(((double)q + 0.5) / 255.0);
}

/* ------------------------------------------------------------------------- */
/* Node passes                                                                 */
/* ------------------------------------------------------------------------- */

void rfo_passthrough(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                     int W, int H, int fmt)
{
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            store_px(out, out_pitch, fmt, x, y, load_px(in, in_pitch, fmt, x, y));
}

void rfo_gaussian(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                  int W, int H, int fmt, int radius, const float* w)
{
    /* f32 intermediate of the H pass: never re-quantised inside the node */
    float* tmp = (float*)malloc((size_t)W * (size_t)H * 4 * sizeof(float));
    if (!tmp) abort();

    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int i = -radius; i <= radius; ++i) {
                px4 p = load_px(in, in_pitch, fmt, clampi(x + i, 0, W - 1), y);
                float wi = w[i < 0 ? -i : i];
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(wi, p.c[k], acc[k]);
            }
            memcpy(tmp + ((size_t)y * W + x) * 4, acc, 16);
        }
    }

    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            px4 o = {{0.0f, 0.0f, 0.0f, 0.0f}};
            for (int j = -radius; j <= radius; ++j) {
                const float* t = tmp + ((size_t)clampi(y + j, 0, H - 1) * W + x) * 4;
                float wj = w[j < 0 ? -j : j];
                for (int k = 0; k < 4; ++k) o.c[k] = fmaf(wj, t[k], o.c[k]);
            }
            store_px(out, out_pitch, fmt, x, y, o);
        }
    }
    free(tmp);
}

static inline float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

void rfo_colour_grade(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                      int W, int H, int fmt, float slope, float offset, float saturation)
{
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            px4 p = load_px(in, in_pitch, fmt, x, y);
            float tr = fmaf(p.c[0], slope, offset);
            float tg = fmaf(p.c[1], slope, offset);
            float tb = fmaf(p.c[2], slope, offset);
            /* Rec.709 luma */
            float luma = fmaf(0.0722f, tb, fmaf(0.7152f, tg, 0.2126f * tr));
            px4 o;
            o.c[0] = clamp01(fmaf(saturation, tr - luma, luma));
            o.c[1] = clamp01(fmaf(saturation, tg - luma, luma));
            o.c[2] = clamp01(fmaf(saturation, tb - luma, luma));
            o.c[3] = p.c[3];
            store_px(out, out_pitch, fmt, x, y, o);
        }
    }
}

void rfo_sharpen(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                 int W, int H, int fmt, float amount)
{
    float wc, ws;
    rfo_sharpen_weights(amount, &wc, &ws);

    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            int xm = clampi(x - 1, 0, W - 1), xp = clampi(x + 1, 0, W - 1);
            int ym = clampi(y - 1, 0, H - 1), yp = clampi(y + 1, 0, H - 1);
            /* ascending tap order, y outer, x inner; zero taps skipped */
            px4 n = load_px(in, in_pitch, fmt, x,  ym);
            px4 l = load_px(in, in_pitch, fmt, xm, y);
            px4 c = load_px(in, in_pitch, fmt, x,  y);
            px4 r = load_px(in, in_pitch, fmt, xp, y);
            px4 s = load_px(in, in_pitch, fmt, x,  yp);
            px4 o;
            for (int k = 0; k < 4; ++k) {
                float acc = 0.0f;
                acc = fmaf(ws, n.c[k], acc);
                acc = fmaf(ws, l.c[k], acc);
                acc = fmaf(wc, c.c[k], acc);
                acc = fmaf(ws, r.c[k], acc);
                acc = fmaf(ws, s.c[k], acc);
                o.c[k] = acc;
            }
            store_px(out, out_pitch, fmt, x, y, o);
        }
    }
}

void rfo_conv2d(const void* in, size_t in_pitch, void* out, size_t out_pitch,
                int W, int H, int fmt, int K, const float* weights)
{
    int r = K / 2;
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            px4 o = {{0.0f, 0.0f, 0.0f, 0.0f}};
            for (int dy = -r; dy <= r; ++dy) {
                int yy = clampi(y + dy, 0, H - 1);
                for (int dx = -r; dx <= r; ++dx) {
                    px4 p = load_px(in, in_pitch, fmt, clampi(x + dx, 0, W - 1), yy);
                    float wt = weights[(dy + r) * K + (dx + r)];
                    for (int k = 0; k < 4; ++k) o.c[k] = fmaf(wt, p.c[k], o.c[k]);
                }
            }
            store_px(out, out_pitch, fmt, x, y, o);
        }
    }
}

void rfo_mix(const void* a, size_t a_pitch, const void* b, size_t b_pitch,
             void* out, size_t out_pitch, int W, int H, int fmt, float mix)
{
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            px4 pa = load_px(a, a_pitch, fmt, x, y);
            px4 pb = load_px(b, b_pitch, fmt, x, y);
            px4 o;
            for (int k = 0; k < 4; ++k) o.c[k] = fmaf(mix, pb.c[k] - pa.c[k], pa.c[k]);
            store_px(out, out_pitch, fmt, x, y, o);
        }
    }
}

/* split_luma (authored; a node with TWO output images -- the reference allocates one image per output binding,
 * pipeline_graph.rs:205-224): luma = fma(0.0722, b, fma(0.7152, g, 0.2126 r)); luma_image = (l, l, l, a);
 * chroma_image = (fma(0.5, c - l, 0.5) for r, g, b; a).  A NULL output is not written. */
void rfo_split_luma(const void* in, size_t in_pitch, void* luma, size_t luma_pitch, void* chroma, size_t chroma_pitch,
                    int W, int H, int fmt)
{
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            px4 p = load_px(in, in_pitch, fmt, x, y);
            const float l = fmaf(0.0722f, p.c[2], fmaf(0.7152f, p.c[1], 0.2126f * p.c[0]));
            px4 a = {{l, l, l, p.c[3]}};
            px4 b = {{fmaf(0.5f, p.c[0] - l, 0.5f), fmaf(0.5f, p.c[1] - l, 0.5f), fmaf(0.5f, p.c[2] - l, 0.5f), p.c[3]}};
            if (luma) store_px(luma, luma_pitch, fmt, x, y, a);
            if (chroma) store_px(chroma, chroma_pitch, fmt, x, y, b);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* sRGB boundary.  The reference leaves this arithmetic to the Vulkan driver   */
/* (format-converting vkCmdBlitImage, src/vulkan/command.rs:97-141); the       */
/* restatement fixes it as: decode = 256-entry table of the exact EOTF;        */
/* encode = count of code thresholds <= v (monotone, transcendental-free).     */
/* Alpha is linear on both sides.                                              */
/* ------------------------------------------------------------------------- */

void rfo_upload_srgb8(const uint8_t* rgba, size_t stride, void* img, size_t pitch,
                      int W, int H, int fmt)
{
    float eotf[256], thr[255];
    rfo_srgb_tables(eotf, thr);
    for (int y = 0; y < H; ++y) {
        const uint8_t* src = rgba + (size_t)y * stride;
        for (int x = 0; x < W; ++x) {
            px4 p;
            p.c[0] = eotf[src[x * 4 + 0]];
            p.c[1] = eotf[src[x * 4 + 1]];
            p.c[2] = eotf[src[x * 4 + 2]];
            p.c[3] = unorm8_to_f32(src[x * 4 + 3]);
            store_px(img, pitch, fmt, x, y, p);
        }
    }
}

static inline uint8_t srgb_encode(float v, const float thr[255])
{
    /* number of thresholds <= v; NaN compares false everywhere -> 0 */
    int lo = 0, hi = 255;               /* answer in [lo, hi] */
    while (lo < hi) {
        int mid = (lo + hi) / 2;        /* thr[mid] <= v  <=>  answer > mid */
        if (thr[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return (uint8_t)lo;
}

void rfo_download_srgb8(const void* img, size_t pitch, uint8_t* rgba, size_t stride,
                        int W, int H, int fmt)
{
    float eotf[256], thr[255];
    rfo_srgb_tables(eotf, thr);
    for (int y = 0; y < H; ++y) {
        uint8_t* dst = rgba + (size_t)y * stride;
        for (int x = 0; x < W; ++x) {
            px4 p = load_px(img, pitch, fmt, x, y);
            dst[x * 4 + 0] = srgb_encode(p.c[0], thr);
            dst[x * 4 + 1] = srgb_encode(p.c[1], thr);
            dst[x * 4 + 2] = srgb_encode(p.c[2], thr);
            dst[x * 4 + 3] = f32_to_unorm8(p.c[3]);
        }
    }
}
