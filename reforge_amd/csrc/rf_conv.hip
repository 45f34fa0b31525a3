// rf_conv.hip -- dense KxK convolution kernels (conv2d node) for gfx950: a 16x16 LDS-tile
// VALU kernel for small K, the banded contraction on v_mfma_f32_16x16x4_f32 (default from
// 9x9), and a register-blocked VALU kernel.  All three are bit-identical to the oracle's
// (dy outer, dx inner) fmaf chain.  See DESIGN.md section 6.2.
#include "rf_device.h"

namespace rf {

// ---------------------------------------------------------------------------------
// conv2d: dense KxK correlation on a 16x16 output tile with an LDS halo tile.
// First (VALU) version: one output texel per thread, taps from LDS in the oracle's
// order.  The MFMA/Toeplitz formulation is the planned replacement.
// ---------------------------------------------------------------------------------
template <class Px>
__global__ __launch_bounds__(256) void conv2d_tile_kernel(const char* src, size_t src_pitch, char* dst, size_t dst_pitch,
                                                          int W, int row_lo, int row_hi, int y0, int y1, int K,
                                                          const float* __restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    f4* tile = reinterpret_cast<f4*>(dyn_smem);
    const int r = K / 2;
    const int TW = 16 + 2 * r;
    float* wl = reinterpret_cast<float*>(tile + TW * TW);

    const int tx = (int)(threadIdx.x & 15), ty = (int)(threadIdx.x >> 4);
    const int bx = (int)blockIdx.x * 16, by = y0 + (int)blockIdx.y * 16;

    for (int i = (int)threadIdx.x; i < K * K; i += 256) wl[i] = weights[i];
    for (int i = (int)threadIdx.x; i < TW * TW; i += 256) {
        int lx = i % TW, ly = i / TW;
        int gx = min(max(bx + lx - r, 0), W - 1);
        int gy = min(max(by + ly - r, row_lo), row_hi);
        tile[i] = Px::decode(Px::load(src + (ptrdiff_t)gy * (ptrdiff_t)src_pitch, (unsigned)gx * (unsigned)Px::BPP));
    }
    __syncthreads();

    f4 acc = f4_zero();
    for (int dy = 0; dy < K; ++dy) {
        const f4* trow = tile + (ty + dy) * TW + tx;
        const float* wrow = wl + dy * K;
        for (int dx = 0; dx < K; ++dx) acc = fma4(wrow[dx], trow[dx], acc);
    }
    const int ox = bx + tx, oy = by + ty;
    if (ox < W && oy < y1) Px::store(dst + (ptrdiff_t)oy * (ptrdiff_t)dst_pitch, (unsigned)ox * (unsigned)Px::BPP, acc);
}

// ---------------------------------------------------------------------------------
// conv2d on the matrix cores: dense KxK correlation as a banded (Toeplitz) contraction on
// v_mfma_f32_16x16x4_f32.  This is the im2col idea restricted to what a single shared
// KxK kernel allows: the "patch matrix" has only ONE filter column, so instead the
// horizontal taps of one weight row become a banded matrix
//     T_dy[x_in][x_out] = w[dy][x_in - x_out]   (0 outside the K taps)
// and for every weight row dy
//     Out[(y,c)][x_out] += In[(y+dy, c)][x_in] * T_dy[x_in][x_out]
// with M = 16 = 4 output rows x 4 channels, N = 16 output columns, K-dim = the 16+2r
// input columns (padded to a multiple of 4).  31x31: 12 MFMAs per weight row per tile,
// 31/48 = 65 % of the multiply-adds are real taps.
//
// Exactness: an f32 MFMA is a k-ordered chain of single-rounding fmaf (MI355X guide,
// "FP32-input MFMA"), the contraction index runs over x_in ascending = dx ascending, weight
// rows are accumulated dy ascending, and a zero band entry adds exactly nothing to a finite
// sum -- so the result is bit-identical to the oracle's (dy outer, dx inner) fmaf chain for
// finite inputs.  (A non-finite texel poisons the whole 16-column tile row it feeds instead
// of only the K columns around it: 0 * inf = NaN.)
//
// Data movement: one workgroup (4 waves = 64 output columns) walks DOWN a chunk of rows 8
// output rows at a time, keeping the 8+2r input rows it needs in an LDS ring (row pitch = 8
// mod 32 dwords so the 16 (row,channel) x 2 (k) operand reads of a lane group hit 32 banks);
// each step loads only the 8 new rows, so an input row is fetched once per strip.
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kConvStripW = 64;     // output columns per workgroup
constexpr int kConvStepRows = 8;    // output rows per step (2 M-tiles of 4 rows per wave)
constexpr int kConvWRow = 64;       // dwords per padded weight row: 15 zeros, K taps, zeros

static int conv_mfma_pitch(int r) { return (((kConvStripW + 2 * r) * 4 + 31) & ~31) + 8; }   // dwords, = 8 mod 32
static int conv_mfma_ring(int r) { return (kConvStepRows + 2 * r + 3) & ~3; }

template <class Px, int STEPS>   // STEPS = MFMA k-steps per weight row = ceil((16 + 2r) / 4), compile-time so the row unrolls
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(const char* src, size_t src_pitch, char* dst, size_t dst_pitch,
                                                          int W, int row_lo, int row_hi, int y0, int y1, int rows_per_chunk,
                                                          int K, int pitch, int ring, const float* __restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    float* wpad = reinterpret_cast<float*>(dyn_smem);            // [K][64]
    float* tile = wpad + K * kConvWRow;                           // [ring][pitch]
    const int r = K / 2;
    const int xin = kConvStripW + 2 * r;                          // input columns of the strip
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kq = lane >> 4;                      // B: column j, k index kq;  A: row i = lane&15, k index kq
    const int a_yy = (lane & 15) >> 2, a_c = lane & 3;

    const int x_out0 = (int)blockIdx.x * kConvStripW;
    const int cy0 = y0 + (int)blockIdx.y * rows_per_chunk;
    const int cy1 = min(cy0 + rows_per_chunk, y1);
    if (cy0 >= cy1) return;

    // padded weight rows: wpad[dy][15 + t] = w[dy][t]
    for (int i = tid; i < K * kConvWRow; i += 256) {
        int dy = i / kConvWRow, t = i % kConvWRow - 15;
        wpad[i] = (t >= 0 && t < K) ? weights[dy * K + t] : 0.0f;
    }

    // columns beyond the strip's last input column are multiplied by zero band entries: they
    // must be finite, so the whole ring starts as zeros
    for (int i = tid; i < ring * pitch; i += 256) tile[i] = 0.0f;
    __syncthreads();

    const int first_in = cy0 - r;                                 // frame row held by ring offset 0
    int loaded_to = first_in;                                     // rows [first_in, loaded_to) are in the ring
    for (int ys = cy0; ys < cy1; ys += kConvStepRows) {
        // stage the rows this step needs that are not in the ring yet (clamp-to-edge on load)
        const int need_to = ys + kConvStepRows + r;
        const int nrows = need_to - loaded_to;
        for (int i = tid; i < nrows * xin; i += 256) {
            const int rr = loaded_to + i / xin, xx = i % xin;
            const int gy = min(max(rr, row_lo), row_hi);
            const int gx = min(max(x_out0 - r + xx, 0), W - 1);
            const f4 v = Px::decode(Px::load(src + (ptrdiff_t)gy * (ptrdiff_t)src_pitch, (unsigned)gx * (unsigned)Px::BPP));
            const int slot = (rr - first_in) % ring;
            *reinterpret_cast<f4*>(tile + slot * pitch + xx * 4) = v;
        }
        loaded_to = need_to;
        __syncthreads();

        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // ring slot of input row (ys + 4m + yy + dy - r) = ((ys - cy0) + 4m + yy + dy) mod ring
        int slot0 = ((ys - cy0) + a_yy) % ring;
        int slot1 = ((ys - cy0) + 4 + a_yy) % ring;
        const int a_col = (16 * wave + kq) * 4 + a_c;              // dword offset of (column 16w+k, channel c)
        const float* bptr = wpad + 15 + kq - j;
        for (int dy = 0; dy < K; ++dy) {
            const float* a0 = tile + slot0 * pitch + a_col;
            const float* a1 = tile + slot1 * pitch + a_col;
            const float* b = bptr + dy * kConvWRow;
            // all operand reads of the weight row are issued ahead of its MFMAs
            float bv[STEPS], av0[STEPS], av1[STEPS];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                bv[s] = b[4 * s];
                av0[s] = a0[16 * s];
                av1[s] = a1[16 * s];
            }
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[s], bv[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[s], bv[s], acc1, 0, 0, 0);
            }
            slot0 = slot0 + 1 == ring ? 0 : slot0 + 1;
            slot1 = slot1 + 1 == ring ? 0 : slot1 + 1;
        }
        // D: column j = lane&15, row i = 4*(lane>>4) + reg  =>  output row yy = lane>>4, channel = reg
        const int ox = x_out0 + 16 * wave + j;
        const int oy0 = ys + kq, oy1 = ys + 4 + kq;
        if (ox < W) {
            if (oy0 < cy1) Px::store(dst + (ptrdiff_t)oy0 * (ptrdiff_t)dst_pitch, (unsigned)ox * (unsigned)Px::BPP, make_float4(acc0[0], acc0[1], acc0[2], acc0[3]));
            if (oy1 < cy1) Px::store(dst + (ptrdiff_t)oy1 * (ptrdiff_t)dst_pitch, (unsigned)ox * (unsigned)Px::BPP, make_float4(acc1[0], acc1[1], acc1[2], acc1[3]));
        }
        __syncthreads();      // the next step overwrites the oldest rows
    }
}

// ---------------------------------------------------------------------------------
// conv2d, register-blocked VALU formulation.  On gfx950 an f32 MFMA runs at the f32 VECTOR
// rate (MI355X guide: 64 FLOP/clk/SIMD either way), so the banded MFMA contraction above
// pays for its zero band entries (35 % at 31x31) with nothing in return; this kernel does
// only the real taps.  Each lane accumulates 4 consecutive output columns of one row
// (4 f4 accumulators) and slides a register window along the row: one ds_read_b128 per
// 8 v_pk_fma_f32.  Tap order is exactly the oracle's (dy outer, dx inner).
//
// Workgroup = 8 waves = 128 output columns x 16 rows per step (lane = 32 column groups x
// 2 rows; two waves per SIMD, which the VALU needs to issue every other cycle), walking down a
// chunk with the 16+2r input rows in an LDS ring, like the MFMA kernel.  LDS row layout:
// column c lives at sub-row (c & 3), position (c >> 2), so the 32 lanes of a row read 32
// CONSECUTIVE texels for any tap (lane lx reads column 4*lx + m: the sub-row m & 3 is the same
// in every lane); the row pitch is a multiple of 256 B.
// ---------------------------------------------------------------------------------
constexpr int kCvStripW = 128;      // output columns per workgroup
constexpr int kCvStepRows = 16;     // output rows per step
constexpr int kCvT = 4;             // output columns per lane
constexpr int kCvSub = 40;          // texels per sub-row: (128 + 30 + 2) / 4; 4 x 40 x 16 B = 2560 B = 10 x 256 B per row

template <class Px, int K>   // K is compile-time: the tap loop unrolls completely, the register window rotates by renaming
__global__ __launch_bounds__(512) void conv2d_valu_kernel(const char* src, size_t src_pitch, char* dst, size_t dst_pitch,
                                                          int W, int row_lo, int row_hi, int y0, int y1, int rows_per_chunk,
                                                          int ring, const float* __restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    constexpr int KQ = (K + 3) / 4;                               // f4 per padded weight row
    constexpr int WN = 8;                                         // register window: 4 texels in use + 4 in flight
    f4* wl = reinterpret_cast<f4*>(dyn_smem);                    // weights, [K][KQ] f4
    f4* tile = wl + K * KQ;                                       // [ring][4][kCvSub]
    constexpr int kRowTexels = kCvT * kCvSub;                     // 160 texels = 2560 B per ring row
    constexpr int r = K / 2;
    constexpr int xin = kCvStripW + 2 * r;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 31, ly = lane >> 5;

    const int x_out0 = (int)blockIdx.x * kCvStripW;
    const int cy0 = y0 + (int)blockIdx.y * rows_per_chunk;
    const int cy1 = min(cy0 + rows_per_chunk, y1);
    if (cy0 >= cy1) return;

    // weight rows into LDS (a per-tap scalar load from global would stall the wave on every tap)
    for (int i = tid; i < K * KQ * 4; i += 512) {
        const int dy = i / (KQ * 4), dx = i % (KQ * 4);
        reinterpret_cast<float*>(wl)[i] = dx < K ? weights[dy * K + dx] : 0.0f;
    }
    const int first_in = cy0 - r;
    int loaded_to = first_in;
    for (int ys = cy0; ys < cy1; ys += kCvStepRows) {
        const int need_to = ys + kCvStepRows + r;
        const int nrows = need_to - loaded_to;
        for (int i = tid; i < nrows * xin; i += 512) {
            const int rr = loaded_to + i / xin, c = i % xin;
            const int gy = min(max(rr, row_lo), row_hi);
            const int gx = min(max(x_out0 - r + c, 0), W - 1);
            const f4 v = Px::decode(Px::load(src + (ptrdiff_t)gy * (ptrdiff_t)src_pitch, (unsigned)gx * (unsigned)Px::BPP));
            const int slot = (rr - first_in) % ring;
            tile[slot * kRowTexels + (c & 3) * kCvSub + (c >> 2)] = v;
        }
        loaded_to = need_to;
        __syncthreads();

        f4 acc[kCvT];
#pragma unroll
        for (int t = 0; t < kCvT; ++t) acc[t] = f4_zero();
        int slot = ((ys - cy0) + 2 * wave + ly) % ring;           // ring slot of input row (ys + 2*wave + ly + dy - r)
        for (int dy = 0; dy < K; ++dy) {
            const f4* row = tile + slot * kRowTexels + lx;        // texel m of this lane's window: row[(m & 3) * kCvSub + (m >> 2)]
            const f4* wrow = wl + dy * KQ;                        // same address in every lane: LDS broadcast
            float wv[KQ * 4];
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const f4 q = wrow[i];
                wv[4 * i] = q.x; wv[4 * i + 1] = q.y; wv[4 * i + 2] = q.z; wv[4 * i + 3] = q.w;
            }
            f4 win[WN];
#pragma unroll
            for (int m = 0; m < WN; ++m) win[m] = row[(m & 3) * kCvSub + (m >> 2)];
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
#pragma unroll
                for (int t = 0; t < kCvT; ++t) acc[t] = fma4(wv[dx], win[(dx + t) % WN], acc[t]);
                // texel dx is done: its register takes texel dx + WN (needed 4 taps from now)
                if (dx + WN <= K - 1 + kCvT - 1) {
                    const int m = dx + WN;
                    win[dx % WN] = row[(m & 3) * kCvSub + (m >> 2)];
                }
            }
            slot = slot + 1 == ring ? 0 : slot + 1;
        }
        const int oy = ys + 2 * wave + ly;
        if (oy < cy1) {
            char* orow = dst + (ptrdiff_t)oy * (ptrdiff_t)dst_pitch;
#pragma unroll
            for (int t = 0; t < kCvT; ++t) {
                const int ox = x_out0 + kCvT * lx + t;
                if (ox < W) Px::store(orow, (unsigned)ox * (unsigned)Px::BPP, acc[t]);
            }
        }
        __syncthreads();
    }
}

template <class Px, int K = 9>
static hipError_t launch_conv_valu(int k, dim3 grid, size_t lds, hipStream_t stream, const char* src, size_t src_pitch, char* dst,
                                   size_t dst_pitch, int W, int row_lo, int row_hi, int y0, int y1, int rpc, int ring, const float* weights)
{
    if constexpr (K > 2 * kMaxRadius + 1) {
        return hipErrorInvalidValue;
    } else {
        if (k != K) return launch_conv_valu<Px, K + 2>(k, grid, lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc, ring, weights);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_valu_kernel<Px, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv2d_valu_kernel<Px, K>), grid, dim3(512), lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc,
                           ring, weights);
        return hipGetLastError();
    }
}

// k-steps 5..12 cover 9x9 (r = 4 -> 6) .. 31x31 (r = 15 -> 12); smaller kernels use the tile kernel
template <class Px, int STEPS = 5>
static hipError_t launch_conv_mfma(int steps, dim3 grid, size_t lds, hipStream_t stream, const char* src, size_t src_pitch, char* dst,
                                   size_t dst_pitch, int W, int row_lo, int row_hi, int y0, int y1, int rpc, int K, int pitch, int ring,
                                   const float* weights)
{
    if constexpr (STEPS > 12) {
        return hipErrorInvalidValue;
    } else {
        if (steps != STEPS)
            return launch_conv_mfma<Px, STEPS + 1>(steps, grid, lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc, K,
                                                   pitch, ring, weights);
        static bool attr_set = false;   // more than 64 KiB of dynamic LDS needs the attribute
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<Px, STEPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv2d_mfma_kernel<Px, STEPS>), grid, dim3(256), lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0,
                           y1, rpc, K, pitch, ring, weights);
        return hipGetLastError();
    }
}

// Rows per chunk for the conv kernels: the chunk count is chosen so that strips x chunks fills
// whole rounds of the `slots` workgroups the chip holds at once (a 60-strip frame cut into 13
// chunks runs 780 workgroups = 3.05 rounds of 256 and wastes a quarter of the last one).
static int conv_rows_per_chunk(int rows, int strips, int slots, int step, int min_rows)
{
    int best_c = 1;
    double best_eff = 0.0;
    for (int rounds = 2; rounds <= 8; ++rounds) {
        int c = rounds * slots / strips;
        if (c < 1) c = 1;
        int rpc = ((rows + c - 1) / c + step - 1) / step * step;
        if (rpc < min_rows) continue;
        const int wgs = strips * ((rows + rpc - 1) / rpc);
        const double eff = (double)wgs / (double)((wgs + slots - 1) / slots * slots);
        if (eff > best_eff + 0.02) { best_eff = eff; best_c = c; }
    }
    int rpc = ((rows + best_c - 1) / best_c + step - 1) / step * step;
    return rpc < min_rows ? min_rows : rpc;
}

template <class Px>
static hipError_t launch_conv2d_px(const Op& op, Image src, Image dst, const Geom& g, const StreamTuning& tune, hipStream_t stream)
{
    const int K = 2 * op.radius + 1;
    if (op.radius < 0 || op.radius > kMaxRadius || !op.dev_weights) return hipErrorInvalidValue;
    const int rows = g.y1 - g.y0;
    if (rows <= 0 || g.W <= 0) return hipSuccess;
    // large kernels run on the matrix cores; small ones keep the 16x16 LDS-tile kernel
    const int rows_cv = g.y1 - g.y0;
    if (tune.conv_path == 3 && K >= 9 && rows_cv > 0 && g.W > 0) {
        const int ring = (kCvStepRows + 2 * op.radius + 3) & ~3;
        const size_t lds = ((size_t)ring * kCvT * kCvSub + (size_t)K * ((K + 3) / 4)) * sizeof(f4);
        const int strips = (g.W + kCvStripW - 1) / kCvStripW;
        int rpc = conv_rows_per_chunk(rows_cv, strips, 256, kCvStepRows, 2 * kCvStepRows);   // one 124 KiB workgroup per CU
        if (tune.rows_per_chunk > 0) rpc = (tune.rows_per_chunk + kCvStepRows - 1) / kCvStepRows * kCvStepRows;
        dim3 grid((unsigned)strips, (unsigned)((rows_cv + rpc - 1) / rpc));
        return launch_conv_valu<Px>(K, grid, lds, stream, static_cast<const char*>(src.base), src.pitch, static_cast<char*>(dst.base),
                                    dst.pitch, g.W, g.row_lo, g.row_hi, g.y0, g.y1, rpc, ring, op.dev_weights);
    }
    const bool mfma = tune.conv_path == 2 || (tune.conv_path != 1 && K >= 9);
    if (mfma) {
        const int pitch = conv_mfma_pitch(op.radius), ring = conv_mfma_ring(op.radius);
        const size_t lds = ((size_t)K * kConvWRow + (size_t)ring * pitch) * sizeof(float);
        const int strips = (g.W + kConvStripW - 1) / kConvStripW;
        // ~4 workgroups per CU in flight; chunks are whole steps of 8 rows
        int rpc = conv_rows_per_chunk(rows, strips, 512, kConvStepRows, 4 * kConvStepRows);   // two 70 KiB workgroups per CU
        if (tune.rows_per_chunk > 0) rpc = (tune.rows_per_chunk + kConvStepRows - 1) / kConvStepRows * kConvStepRows;
        dim3 grid((unsigned)strips, (unsigned)((rows + rpc - 1) / rpc));
        return launch_conv_mfma<Px>((16 + 2 * op.radius + 3) / 4, grid, lds, stream, static_cast<const char*>(src.base), src.pitch,
                                    static_cast<char*>(dst.base), dst.pitch, g.W, g.row_lo, g.row_hi, g.y0, g.y1, rpc, K, pitch,
                                    ring, op.dev_weights);
    }
    const int TW = 16 + 2 * op.radius;
    size_t lds = (size_t)TW * TW * sizeof(f4) + (size_t)K * K * sizeof(float);
    dim3 grid((unsigned)((g.W + 15) / 16), (unsigned)((rows + 15) / 16));
    hipLaunchKernelGGL((conv2d_tile_kernel<Px>), grid, dim3(256), lds, stream,
                       static_cast<const char*>(src.base), src.pitch, static_cast<char*>(dst.base), dst.pitch,
                       g.W, g.row_lo, g.row_hi, g.y0, g.y1, K, op.dev_weights);
    return hipGetLastError();
}

hipError_t launch_conv2d(int fmt, const Op& op, Image src, Image dst, const Geom& g, const StreamTuning& tune, hipStream_t stream)
{
    if (fmt == kFmtRGBA8) return launch_conv2d_px<PxU8>(op, src, dst, g, tune, stream);
    if (fmt == kFmtRGBA32F) return launch_conv2d_px<PxF32>(op, src, dst, g, tune, stream);
    return hipErrorInvalidValue;
}

}  // namespace rf
