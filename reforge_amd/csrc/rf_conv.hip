// rf_conv.hip -- dense KxK convolution kernels (conv2d node) for gfx950: the register-blocked
// VALU kernel (default), the banded contraction on v_mfma_f32_16x16x4_f32 (RF_CONV_PATH=2,
// K >= 9) and a plain 16x16 LDS-tile kernel (RF_CONV_PATH=1; K = 1).  All three are
// bit-identical to the oracle's (dy outer, dx inner) fmaf chain.  See DESIGN.md section 6.2.
#include <type_traits>

#include "rf_device.h"

namespace rf {

// ---------------------------------------------------------------------------------
// conv2d: dense KxK correlation on a 16x16 output tile with an LDS halo tile.
// One output texel per thread, taps from LDS in the oracle's order: the simple baseline the
// other two kernels are measured against (8.4 ms at 31x31 8K).
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
// Data movement: one workgroup (4 waves = 64 output columns) walks DOWN a chunk of rows 16
// output rows at a time, keeping the 16+2r input rows it needs in an LDS ring; each step loads
// only the 16 new rows, so an input row is fetched once per strip.  Software pipelining:
//  * the ring is stored TRANSPOSED: a row is 16 sub-rows (column & 3, channel) of 24 dwords,
//    so the A operands of four consecutive k-steps of a lane are 4 consecutive dwords: one
//    ds_read_b128 instead of four ds_read_b32 (3 per M-tile per weight row at 31x31);
//  * a wave owns 4 M-tiles (16 output rows x 16 columns): the B operand (band of the weight
//    row) is read once for 4 tiles;
//  * operands of weight row dy+1 are read into a second register set while the 48 MFMAs of
//    row dy issue (sched_group_barrier pins the interleave: 1 LDS read, 2 MFMAs, ...) --
//    left to itself hipcc sinks each operand read next to its MFMA and every group of 4 MFMAs
//    waits a full LDS latency (the first version of this kernel: MFMA pipe 76 % busy, this
//    one 84 %, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES);
//  * the 16 input rows of the NEXT step are fetched from global into registers before the
//    weight-row loop and written to the ring after it.
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxDevices = 64;
static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

constexpr int kConvStripW = 64;     // output columns per workgroup
constexpr int kConvWRow = 64;       // dwords per padded weight row: 15 zeros, K taps, zeros
constexpr int kCm2StepRows = 16;                 // output rows per step: 4 M-tiles of 4 rows per wave
constexpr int kCm2NQ = 24;                       // dwords per sub-row: (64 + 30 + 2) / 4 columns
constexpr int kCm2Pitch = 16 * kCm2NQ + 4;       // dwords per ring row (4 mod 32: the rows of a lane group spread over the banks)

template <class Px, int STEPS>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(const char* src, size_t src_pitch, char* dst, size_t dst_pitch,
                                                           int W, int row_lo, int row_hi, int y0, int y1, int rows_per_chunk,
                                                           int K, const float* __restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    constexpr int G = (STEPS + 3) / 4;                            // ds_read_b128 per M-tile per weight row
    constexpr int MT = kCm2StepRows / 4;                          // M-tiles per wave
    float* wpad = reinterpret_cast<float*>(dyn_smem);            // [K][64]
    float* tile = wpad + K * kConvWRow;                           // [ring][kCm2Pitch]
    const int r = K / 2;
    const int ring = kCm2StepRows + 2 * r;
    const int xin = kConvStripW + 2 * r;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int a_yy = j >> 2, a_c = j & 3;

    const int x_out0 = (int)blockIdx.x * kConvStripW;
    const int cy0 = y0 + (int)blockIdx.y * rows_per_chunk;
    const int cy1 = min(cy0 + rows_per_chunk, y1);
    if (cy0 >= cy1) return;

    for (int i = tid; i < K * kConvWRow; i += 256) {
        int dy = i / kConvWRow, t = i % kConvWRow - 15;
        wpad[i] = (t >= 0 && t < K) ? weights[dy * K + t] : 0.0f;
    }
    for (int i = tid; i < ring * kCm2Pitch; i += 256) tile[i] = 0.0f;   // pad columns meet zero band entries: must be finite
    __syncthreads();

    const int first_in = cy0 - r;
    auto fetch = [&](int rr, int xx) -> f4 {
        const int gy = min(max(rr, row_lo), row_hi);
        const int gx = min(max(x_out0 - r + xx, 0), W - 1);
        return Px::decode(Px::load(src + (ptrdiff_t)gy * (ptrdiff_t)src_pitch, (unsigned)gx * (unsigned)Px::BPP));
    };
    auto put = [&](int rr, int xx, const f4& v) {
        float* q = tile + ((rr - first_in) % ring) * kCm2Pitch + ((xx & 3) * 4) * kCm2NQ + (xx >> 2);
        q[0] = v.x;
        q[kCm2NQ] = v.y;
        q[2 * kCm2NQ] = v.z;
        q[3 * kCm2NQ] = v.w;
    };
    // first fill: the ring's worth of rows of step 0
    int loaded_to = cy0 + kCm2StepRows + r;
    for (int i = tid; i < ring * xin; i += 256) put(first_in + i / xin, i % xin, fetch(first_in + i / xin, i % xin));
    __syncthreads();

    constexpr int NPF = (kCm2StepRows * (kConvStripW + 2 * kMaxRadius) + 255) / 256;   // texels per thread of a 16-row refill
    const int a_off = (kq * 4 + a_c) * kCm2NQ + 4 * wave;         // dword offset of this lane's A run inside a ring row
    const float* bbase = wpad + 15 + kq - j;

    for (int ys = cy0; ys < cy1; ys += kCm2StepRows) {
        const bool has_next = ys + kCm2StepRows < cy1;
        f4 pf[NPF];
        if (has_next) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int i = tid + 256 * q;
                if (i < kCm2StepRows * xin) pf[q] = fetch(loaded_to + i / xin, i % xin);
            }
        }

        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 A[2][MT][G];
        float B[2][STEPS];
        int slot[MT];                                              // ring slot the NEXT operand read takes, per M-tile
#pragma unroll
        for (int m = 0; m < MT; ++m) slot[m] = ((ys - cy0) + 4 * m + a_yy) % ring;
        int dy_load = 0;
        auto load = [&](auto bufc) {
            constexpr int buf = decltype(bufc)::value;
            const float* b = bbase + dy_load * kConvWRow;
#pragma unroll
            for (int s2 = 0; s2 < STEPS; ++s2) B[buf][s2] = b[4 * s2];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4* a = reinterpret_cast<const f32x4*>(tile + slot[m] * kCm2Pitch + a_off);
#pragma unroll
                for (int g = 0; g < G; ++g) A[buf][m][g] = a[g];
                slot[m] = slot[m] + 1 == ring ? 0 : slot[m] + 1;
            }
            ++dy_load;
        };
        auto mma = [&](auto bufc) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int s2 = 0; s2 < STEPS; ++s2) {
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[buf][m][s2 >> 2][s2 & 3], B[buf][s2], acc[m], 0, 0, 0);
            }
        };
        auto interleave = [&]() {   // 1 LDS read, 2 MFMAs, ... : the reads of the next row ride in the shadow of this row's MFMAs
#pragma unroll
            for (int q = 0; q < MT * G + (STEPS + 1) / 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        };
        typedef std::integral_constant<int, 0> B0;
        typedef std::integral_constant<int, 1> B1;
        load(B0{});
        for (int dy = 0; dy + 2 < K; dy += 2) {
            load(B1{});
            mma(B0{});
            interleave();
            load(B0{});
            mma(B1{});
            interleave();
        }
        mma(B0{});                                                 // K is odd: the last weight row sits in set 0

        // D: column j = lane & 15, row 4*(lane >> 4) + reg  =>  output row kq of the M-tile, channel = reg
        const int ox = x_out0 + 16 * wave + j;
        if (ox < W) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int oy = ys + 4 * m + kq;
                if (oy < cy1) Px::store(dst + (ptrdiff_t)oy * (ptrdiff_t)dst_pitch, (unsigned)ox * (unsigned)Px::BPP, make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
            }
        }
        __syncthreads();      // every wave is done with the 16 oldest rows
        if (has_next) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int i = tid + 256 * q;
                if (i < kCm2StepRows * xin) put(loaded_to + i / xin, i % xin, pf[q]);
            }
            loaded_to += kCm2StepRows;
        }
        __syncthreads();
    }
}

template <class Px, int STEPS = 5>
static hipError_t launch_conv_mfma(int steps, dim3 grid, size_t lds, hipStream_t stream, const char* src, size_t src_pitch, char* dst,
                                    size_t dst_pitch, int W, int row_lo, int row_hi, int y0, int y1, int rpc, int K, const float* weights)
{
    if constexpr (STEPS > 12) {
        return hipErrorInvalidValue;
    } else {
        if (steps != STEPS)
            return launch_conv_mfma<Px, STEPS + 1>(steps, grid, lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc, K, weights);
        static bool attr_done[kMaxDevices] = {};   // the attribute is per device
        bool& attr_set = attr_done[current_device()];
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_mfma_kernel<Px, STEPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv2d_mfma_kernel<Px, STEPS>), grid, dim3(256), lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0,
                           y1, rpc, K, weights);
        return hipGetLastError();
    }
}

// ---------------------------------------------------------------------------------
// conv2d, register-blocked VALU formulation.  On gfx950 an f32 MFMA runs at the f32 VECTOR
// rate (MI355X guide: 64 FLOP/clk/SIMD either way), so the banded MFMA contraction above
// pays for its zero band entries (35 % at 31x31) with nothing in return; this kernel does
// only the real taps.  Each lane accumulates 8 consecutive output columns of one row
// (8 f4 accumulators) and slides a register window along the row: one ds_read_b128 (1 KiB per
// wave = 8 LDS cycles) per 16 v_pk_fma_f32 (64 issue cycles), so the four SIMDs of a CU together
// keep the LDS pipe half busy -- with 4 columns per lane the kernel was LDS-bound at half the
// VALU rate.  The weights are SCALAR operands (s_load of the row through the scalar cache, see the tap loop): a packed FMA reads two
// VGPR operands and one SGPR pair.  Tap order is exactly the oracle's (dy outer, dx inner).
//
// Workgroup = 8 waves = 128 output columns x 32 rows per step (lane = 16 column groups x
// 4 rows; two waves per SIMD, which the VALU needs to issue every other cycle), walking down a
// chunk with the 32+2r input rows in an LDS ring, like the MFMA kernel.  LDS row layout:
// column c lives at sub-row (c & 7), position (c >> 3), so the 16 lanes of a row read 16
// CONSECUTIVE texels for any tap (lane lx reads column 8*lx + m: the sub-row m & 7 is the same
// in every lane); the row pitch is a multiple of 256 B.
// ---------------------------------------------------------------------------------
constexpr int kCvStripW = 128;      // output columns per workgroup
constexpr int kCvStepRows = 32;     // output rows per step
constexpr int kCvT = 8;             // output columns per lane
constexpr int kCvSub = 20;          // texels per sub-row: (128 + 30 + 2) / 8; 8 x 20 x 16 B = 2560 B = 10 x 256 B per row
constexpr int kCvLanesX = kCvStripW / kCvT;     // 16
constexpr int kCvLanesY = 64 / kCvLanesX;       // 4 rows per wave (of the default <8 columns, 8 waves> shape)
static_assert(kCvLanesY * 8 == 32, "eight waves cover a 32-row step");

// CT output columns per lane, WAVES waves per workgroup: <8, 8> = 16 column groups x 4 rows per wave, two waves per SIMD, is the
// one shape instantiated.  (<4, 16> = 32 column groups x 2 rows per wave, FOUR waves per SIMD on the same 32-row step and the same
// ring, was measured 7 % slower -- profiles/r03_conv_four_waves_probe.txt -- and is not built; conv_path takes 0..3 only.)
template <class Px, int K, int CT = kCvT, int WAVES = 8>   // K is compile-time: the tap loop unrolls completely, the register window rotates by renaming
__global__ __launch_bounds__(64 * WAVES) void conv2d_valu_kernel(const char* src, size_t src_pitch, char* dst, size_t dst_pitch,
                                                          int W, int row_lo, int row_hi, int y0, int y1, int rows_per_chunk,
                                                          int ring, const float* __restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
    constexpr int KQ = (K + 3) / 4;                               // f4 per padded weight row
    constexpr int kLanesX = kCvStripW / CT, kLanesY = 64 / kLanesX, kSub = (kCvT * kCvSub) / CT, THREADS = 64 * WAVES;
    static_assert(kLanesY * WAVES == kCvStepRows, "a step is 32 output rows");
    constexpr int WN = CT + 4;                                  // register window: kCvT texels in use + 4 in flight
    f4* tile = reinterpret_cast<f4*>(dyn_smem);                  // [ring][4][kCvSub]
    constexpr int kRowTexels = kCvT * kCvSub;                     // 160 texels = 2560 B per ring row
    constexpr int r = K / 2;
    constexpr int xin = kCvStripW + 2 * r;
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & (kLanesX - 1), ly = lane / kLanesX;

    const int x_out0 = (int)blockIdx.x * kCvStripW;
    const int cy0 = y0 + (int)blockIdx.y * rows_per_chunk;
    const int cy1 = min(cy0 + rows_per_chunk, y1);
    if (cy0 >= cy1) return;

    const int first_in = cy0 - r;
    auto fetch = [&](int rr, int c) {
        const int gy = min(max(rr, row_lo), row_hi);
        const int gx = min(max(x_out0 - r + c, 0), W - 1);
        return Px::load(src + (ptrdiff_t)gy * (ptrdiff_t)src_pitch, (unsigned)gx * (unsigned)Px::BPP);
    };
    auto put = [&](int rr, int c, typename Px::Raw v) {
        const int slot = (rr - first_in) % ring;
        tile[slot * kRowTexels + (c % CT) * kSub + (c / CT)] = Px::decode(v);
    };
    // first fill: the ring's worth of rows of step 0 (32 + 2r rows)
    int loaded_to = cy0 + kCvStepRows + r;
    for (int i = tid; i < (loaded_to - first_in) * xin; i += THREADS) put(first_in + i / xin, i % xin, fetch(first_in + i / xin, i % xin));
    // the 32 rows of the NEXT step travel global -> registers while this step's 31 weight rows compute, and go to the
    // ring after it (the slots they take are the 32 oldest rows, free once every wave has passed the barrier)
    constexpr int NPF = (kCvStepRows * xin + THREADS - 1) / THREADS;
    for (int ys = cy0; ys < cy1; ys += kCvStepRows) {
        const bool has_next = ys + kCvStepRows < cy1;
        typename Px::Raw pf[NPF];
        __syncthreads();
        if (has_next) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int i = tid + THREADS * q;
                if (i < kCvStepRows * xin) pf[q] = fetch(loaded_to + i / xin, i % xin);
            }
        }

        f4 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[t] = f4_zero();
        int slot = ((ys - cy0) + kLanesY * wave + ly) % ring;   // ring slot of input row (ys + 4*wave + ly + dy - r)
#pragma unroll 1      // one weight row at a time: unrolled (hipcc does it for K <= 7) every row's weights are fetched up front, ~50 SGPRs
                     // and 256 VGPRs with a kilobyte of scratch (7x7 at 4K: 0.63 ms instead of 0.08)
        for (int dy = 0; dy < K; ++dy) {
            const f4* row = tile + slot * kRowTexels + lx;        // texel m of this lane's window: row[(m % T) * kCvSub + m / T]
            // The K weights of this row come through the SCALAR cache into scalar registers (the address is wave-uniform; the
            // constant address space is what lets hipcc pick s_load_dwordx16/x8/..): every v_pk_fma_f32 then takes its weight as
            // an SGPR operand with op_sel_hi broadcasting it to both halves -- one VGPR operand read less per instruction and no
            // LDS broadcast reads.  Round 3: 2.55-2.58 -> 2.32-2.35 ms at 31x31 8K on one box (98.8-100 -> 108.6-109.9 TF), the
            // shader clock under the same package power 2.09 -> 2.23 GHz (profiles/r03_conv_scalar_weights_probe.txt).  Until then
            // the row was read from an LDS copy of the weights: 8 broadcast ds_read_b128 per row and a VGPR per weight.
            float wv[KQ * 4];
            typedef const float __attribute__((address_space(4))) * ScalarWeights;
            const ScalarWeights wrow = (ScalarWeights)(weights + dy * K);
#pragma unroll
            for (int i = 0; i < K; ++i) wv[i] = wrow[i];
            f4 win[WN];
#pragma unroll
            for (int m = 0; m < WN; ++m) win[m] = row[(m % CT) * kSub + m / CT];
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = fma4(wv[dx], win[(dx + t) % WN], acc[t]);
                // texel dx is done: its register takes texel dx + WN (needed 4 taps from now)
                if (dx + WN <= K - 1 + CT - 1) {
                    const int m = dx + WN;
                    win[dx % WN] = row[(m % CT) * kSub + m / CT];
                }
            }
            slot = slot + 1 == ring ? 0 : slot + 1;
        }
        const int oy = ys + kLanesY * wave + ly;
        if (oy < cy1) {
            char* orow = dst + (ptrdiff_t)oy * (ptrdiff_t)dst_pitch;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int ox = x_out0 + CT * lx + t;
                if (ox < W) Px::store(orow, (unsigned)ox * (unsigned)Px::BPP, acc[t]);
            }
        }
        __syncthreads();      // every wave is done with the 32 oldest rows
        if (has_next) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int i = tid + THREADS * q;
                if (i < kCvStepRows * xin) put(loaded_to + i / xin, i % xin, pf[q]);
            }
            loaded_to += kCvStepRows;
        }
    }
}

template <class Px, int CT, int WAVES, int K = 3>
static hipError_t launch_conv_valu(int k, dim3 grid, size_t lds, hipStream_t stream, const char* src, size_t src_pitch, char* dst,
                                   size_t dst_pitch, int W, int row_lo, int row_hi, int y0, int y1, int rpc, int ring, const float* weights)
{
    if constexpr (K > 2 * kMaxRadius + 1) {
        return hipErrorInvalidValue;
    } else {
        if (k != K) return launch_conv_valu<Px, CT, WAVES, K + 2>(k, grid, lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc, ring, weights);
        static bool attr_done[kMaxDevices] = {};   // the attribute is per device
        bool& attr_set = attr_done[current_device()];
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_valu_kernel<Px, K, CT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((conv2d_valu_kernel<Px, K, CT, WAVES>), grid, dim3(64 * WAVES), lds, stream, src, src_pitch, dst, dst_pitch, W, row_lo, row_hi, y0, y1, rpc,
                           ring, weights);
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
    if (best_eff == 0.0) return min_rows;   // small frame: every candidate is shorter than min_rows
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
    // conv_path 0 (auto): the register-blocked VALU kernel from 7x7 (31x31 at 8K: 3.0 ms, the MFMA
    // kernel 3.1 ms; 15x15 at 4K: 0.21 vs 0.30 ms -- the band wastes more the smaller K is), the
    // 16x16 LDS tile below that (3x3 at 4K: 59 us vs 77 us).  1: tile; 2: banded contraction on
    // the matrix cores (K >= 9); 3: VALU.
    const bool mfma = tune.conv_path == 2 && K >= 9;
    const bool valu = !mfma && tune.conv_path != 1 && K >= 3 && (K >= 7 || tune.conv_path != 0);
    if (valu) {
        const int ring = kCvStepRows + 2 * op.radius;
        const size_t lds = (size_t)ring * kCvT * kCvSub * sizeof(f4);
        const int strips = (g.W + kCvStripW - 1) / kCvStripW;
        int rpc = conv_rows_per_chunk(rows, strips, 256, kCvStepRows, 2 * kCvStepRows);   // one workgroup (up to 159 KiB of LDS) per CU
        if (tune.rows_per_chunk > 0) rpc = (tune.rows_per_chunk + kCvStepRows - 1) / kCvStepRows * kCvStepRows;
        dim3 grid((unsigned)strips, (unsigned)((rows + rpc - 1) / rpc));
        // (<4 columns per lane, 16 waves> = four waves per SIMD on the same ring was built and measured in round 3: bit-exact and
        // 7 % SLOWER, 2.60-2.64 ms against 2.43-2.45 at 31x31 8K -- its extra LDS reads pull the clock from 2.11-2.14 down to 1.96 GHz,
        // profiles/r03_conv_four_waves_probe.txt; the kernel keeps the template parameters, the product instantiates <8, 8> only)
        return launch_conv_valu<Px, kCvT, 8>(K, grid, lds, stream, static_cast<const char*>(src.base), src.pitch, static_cast<char*>(dst.base),
                                             dst.pitch, g.W, g.row_lo, g.row_hi, g.y0, g.y1, rpc, ring, op.dev_weights);
    }
    if (mfma) {
        const int ring = kCm2StepRows + 2 * op.radius;
        const size_t lds = ((size_t)K * kConvWRow + (size_t)ring * kCm2Pitch) * sizeof(float);
        const int strips = (g.W + kConvStripW - 1) / kConvStripW;
        int rpc = conv_rows_per_chunk(rows, strips, 512, kCm2StepRows, 2 * kCm2StepRows);   // two 80 KiB workgroups per CU
        if (tune.rows_per_chunk > 0) rpc = (tune.rows_per_chunk + kCm2StepRows - 1) / kCm2StepRows * kCm2StepRows;
        dim3 grid((unsigned)strips, (unsigned)((rows + rpc - 1) / rpc));
        return launch_conv_mfma<Px>((16 + 2 * op.radius + 3) / 4, grid, lds, stream, static_cast<const char*>(src.base), src.pitch,
                                    static_cast<char*>(dst.base), dst.pitch, g.W, g.row_lo, g.row_hi, g.y0, g.y1, rpc, K, op.dev_weights);
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
