// rf_device.h -- device-side basics shared by the gfx950 kernel files: the texel type, the
// packed fma, and the two texel formats (what imageLoad / imageStore do on a storage image,
// shaders/passthrough.comp:9,:12).  Numerics: every multiply-add in the kernels is an explicit
// fmaf in the oracle's order and the files are compiled with -ffp-contract=off, so results are
// bit-identical to oracle/rf_oracle.c for finite inputs.
#pragma once

#ifndef __HIPCC_RTC__   // hiprtc (rf_jit.cpp) supplies the HIP device headers itself and has no libc headers
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "rf_kernels.h"
#endif

#include <type_traits>

namespace rf {

typedef float4 f4;

#define RF_DEV __device__ __forceinline__
// cache-policy modifiers of the stream kernel's row store / LDS-DMA row load (" nt", " sc1", ...): empty in the product.  Measured
// (scripts/mk_variant.sh trees, profiles/r03_cache_policy_probe.txt) and left empty.
#ifndef RF_STORE_MOD
#define RF_STORE_MOD ""
#endif
#ifndef RF_LOAD_MOD
#define RF_LOAD_MOD ""
#endif

RF_DEV f4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// SCALAR BASE OF A VECTOR-MEMORY INSTRUCTION WRITTEN IN ASM.  On gfx9 a vector-memory instruction that reads an SGPR needs five
// wait states behind a VECTOR instruction that wrote it (v_readlane, v_readfirstlane, v_cmp): hipcc's hazard recogniser inserts
// them for its own instructions and cannot see into an asm statement.  A kernel short of scalar registers keeps some in the
// lanes of a VGPR and reloads them with v_readlane right in front of their use -- in front of the asm, as far as hipcc knows --
// and the memory instruction then issues with the STALE register pair: a wild address.  (Round 4 found it as a memory fault of
// the 27- and 31-tap gaussians, the kernels with such reloads, as soon as their walk carried more scalar state; round 3's abort
// of the same kernel, gpurun_out/r03/xcc_tests.log, has the same signature: DESIGN.md 6.1c, profiles/r04_sgpr_hazard_repro.txt.)
// Every such asm therefore copies its base into VCC first and addresses through VCC: a SCALAR read of the register is
// interlocked, a scalar write followed by the memory instruction has no hazard, and VCC costs no register (a pair of the
// kernel's own, tried first, cost the 16384^2 5-stage launch 1.4 %: profiles/r04_static_path_ab.txt).
// tests/test_isa_invariants.py and tests/test_jit_isa.py check every global_* instruction of every kernel for it.
#define RF_SBASE "s_mov_b64 vcc, %[base]\n\t"
// four fmaf as two v_pk_fma_f32 (each lane-pair fma is still one single-rounding fmaf): a VALU
// instruction costs the same issue slot packed or not, and the kernels are issue-sensitive
typedef float v2f __attribute__((ext_vector_type(2)));
// the weight already as a pair {w, w}: stage parameters are stored that way, so that a weight is ONE aligned 8-byte
// scalar load (hipcc otherwise widens neighbouring float loads into overlapping vector loads, which can pin the whole
// parameter block to scratch: it did for the grade -> gaussian chains)
#ifdef RF_EXPERIMENT_NO_FMA      // timing-only build (scripts/mk_variant.sh): the taps are fetched but not summed -- WRONG results
RF_DEV f4 fma4(v2f, f4 v, f4) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); return v; }
RF_DEV f4 fma4(float, f4 v, f4) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); return v; }
#else
RF_DEV f4 fma4(v2f ww, f4 v, f4 a)
{
    const v2f lo = __builtin_elementwise_fma(ww, v2f{v.x, v.y}, v2f{a.x, a.y});
    const v2f hi = __builtin_elementwise_fma(ww, v2f{v.z, v.w}, v2f{a.z, a.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
RF_DEV f4 fma4(float w, f4 v, f4 a)
{
    const v2f ww = {w, w};
    const v2f lo = __builtin_elementwise_fma(ww, v2f{v.x, v.y}, v2f{a.x, a.y});
    const v2f hi = __builtin_elementwise_fma(ww, v2f{v.z, v.w}, v2f{a.z, a.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
#endif

// ---------------------------------------------------------------------------------
// Texel formats: what imageLoad/imageStore do (shaders/passthrough.comp:9,:12)
// ---------------------------------------------------------------------------------
// c / 255 correctly rounded, without the ~12-instruction IEEE division sequence: 1/255 split into a head (the nearest
// float) and a tail, x * tail rounded, then ONE fma x * head + that -- the correctly rounded quotient for all 256 codes
// (tests/test_gpu_parity.py::test_unorm8_decode_all_codes checks every code against the oracle's true division; the
// round-1 form q = x * r, e = fma(-q, 255, x), fma(e, r, q) took three operations).
RF_DEV float unorm8_code_to_f32(float x)
{
    const float hi = 1.0f / 255.0f;                                        // 0x3b808081
    const float lo = (float)(1.0 / 255.0 - (double)(1.0f / 255.0f));       // the part of 1/255 the head misses
    return fmaf(x, hi, x * lo);
}
RF_DEV float unorm8_to_f32(unsigned c) { return unorm8_code_to_f32((float)c); }
// imageStore's float -> UNORM8 as a float: clamp to [0,1] (NaN -> 0), x255, round to nearest even.
// v_med3_f32 returns min3 when an operand is NaN, i.e. 0; the +0 addend of the fma turns a
// clamped -0 into +0, which is what the integer code decodes to.
RF_DEV float unorm8_code(float v)
{
    return rintf(fmaf(__builtin_amdgcn_fmed3f(v, 0.0f, 1.0f), 255.0f, 0.0f));
}
RF_DEV unsigned f32_to_unorm8(float v) { return (unsigned)unorm8_code(v); }
// The same two conversions on a PAIR of channels: the x255 and the /255 run as packed operations (v_pk_fma_f32 /
// v_pk_mul_f32: one instruction for two channels; the same IEEE operation per channel, so the same bits).
RF_DEV v2f unorm8_code2(v2f v)
{
    const v2f c = {__builtin_amdgcn_fmed3f(v.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(v.y, 0.0f, 1.0f)};
    const v2f s = __builtin_elementwise_fma(c, v2f{255.0f, 255.0f}, v2f{0.0f, 0.0f});
    return v2f{rintf(s.x), rintf(s.y)};
}
RF_DEV v2f unorm8_code_to_f32_2(v2f x)
{
    const float hi = 1.0f / 255.0f;
    const float lo = (float)(1.0 / 255.0 - (double)(1.0f / 255.0f));
    return __builtin_elementwise_fma(x, v2f{hi, hi}, x * v2f{lo, lo});
}
struct PxF32 {
    typedef f4 Raw;
    static constexpr int BPP = 16;
    static constexpr bool QUANT = false;
    RF_DEV static Raw load(const char* row, unsigned xoff) { return *reinterpret_cast<const f4*>(row + xoff); }
    RF_DEV static f4 decode(Raw r) { return r; }
    // Take a row out of the prefetch ring into registers of its own.  A real v_mov (the asm
    // is opaque to the optimiser) ends the ring slot's live range HERE, so the refill that
    // follows loads in place and the slot keeps its registers around the loop; without it
    // the compiler renames, copies the slots at the back edge and drains every load in
    // flight with s_waitcnt vmcnt(0) to do so.
    RF_DEV static f4 take(Raw r)
    {
        f4 o;
        asm volatile("v_mov_b32 %0, %1" : "=v"(o.x) : "v"(r.x));
        asm volatile("v_mov_b32 %0, %1" : "=v"(o.y) : "v"(r.y));
        asm volatile("v_mov_b32 %0, %1" : "=v"(o.z) : "v"(r.z));
        asm volatile("v_mov_b32 %0, %1" : "=v"(o.w) : "v"(r.w));
        return o;
    }
    RF_DEV static void store(char* row, unsigned xoff, f4 v)
    {
        // MUST stay one global_store_dwordx4: the stream kernel's counted vmcnt waits rely on
        // one vector-memory instruction per stored row (a non-temporal variant was measured:
        // no gain, and split into four stores it would break the count)
        *reinterpret_cast<f4*>(row + xoff) = v;
    }
    // The same store with a WAVE-UNIFORM row address, written as the scalar-base form (row address in SGPRs, the
    // lane's column offset in one VGPR): the stream kernel's row pointer then advances on the scalar unit, not with a
    // 64-bit vector add per row.
    RF_DEV static void store_row(char* row, unsigned xoff, f4 v)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f d = {v.x, v.y, v.z, v.w};
        // The s_nop is part of the store: on gfx940+ a vector-memory store of more than 8 bytes reads its data registers up to
        // two wait states after it issues, and hipcc's hazard recogniser, which would keep a VALU write to them away, cannot
        // see into an asm statement (scripts/fuzz_graphs.py seeds 7054 / 7063 / 7113: a run-time compiled fork/join kernel
        // scheduled such a write right behind the store -- a few thousand wrong texels per frame, different ones each run).
        asm volatile(RF_SBASE "global_store_dwordx4 %[off], %[data], vcc" RF_STORE_MOD "\n\ts_nop 1" ::[off] "v"(xoff), [data] "v"(d), [base] "s"(row) : "memory", "vcc");
    }
    RF_DEV static f4 requant(f4 v) { return v; }
};

// rgba32f with NON-TEMPORAL row stores: the kernel variant of a launch whose output no launch of the frame reads (the graph's
// result).  Measured (profiles/r03_cache_policy_probe.txt, r03_store_nt_probe.txt: three interleaved rounds on one box): the fused
// 4K chain -1.9 %, gaussian9 at 8K -2..-4 %, the 16384^2 5-stage chain -0..-2 % -- the written rows no longer displace input rows
// neighbouring workgroups still want from L2.  NOT for an image the next launch reads (the unfused 4K chain loses 21 %: its
// intermediate images live in the caches), not for rgba8 (+2..4 %), not on the row LOADS (halo rows are re-used: +15 %).
struct PxF32NT : PxF32 {
    RF_DEV static void store_row(char* row, unsigned xoff, f4 v)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f d = {v.x, v.y, v.z, v.w};
        asm volatile(RF_SBASE "global_store_dwordx4 %[off], %[data], vcc nt\n\ts_nop 1" ::[off] "v"(xoff), [data] "v"(d), [base] "s"(row) : "memory", "vcc");      // the s_nop: see PxF32::store_row
    }
};

struct PxU8 {
    typedef unsigned Raw;
    static constexpr int BPP = 4;
    static constexpr bool QUANT = true;
    RF_DEV static Raw load(const char* row, unsigned xoff) { return *reinterpret_cast<const unsigned*>(row + xoff); }
    RF_DEV static f4 decode(Raw r)
    {
        const v2f lo = unorm8_code_to_f32_2(v2f{(float)(r & 255u), (float)((r >> 8) & 255u)});
        const v2f hi = unorm8_code_to_f32_2(v2f{(float)((r >> 16) & 255u), (float)(r >> 24)});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    RF_DEV static f4 take(Raw r) { return decode(r); }   // the conversion already leaves the ring slot dead
    RF_DEV static unsigned pack(f4 v)
    {
        // v_cvt_pk_u8_f32 IS imageStore's conversion once the value is scaled: it saturates to [0, 255], rounds to
        // nearest even and turns NaN into 0 (scripts/cvt_probe.hip: 1573 values incl. every tie, the neighbours of every
        // tie, infinities and NaN against clamp + rint) -- no clamp and no v_rndne in front of it.
        const v2f lo = __builtin_elementwise_fma(v2f{v.x, v.y}, v2f{255.0f, 255.0f}, v2f{0.0f, 0.0f});
        const v2f hi = __builtin_elementwise_fma(v2f{v.z, v.w}, v2f{255.0f, 255.0f}, v2f{0.0f, 0.0f});
        unsigned o = __builtin_amdgcn_cvt_pk_u8_f32(lo.x, 0u, 0u);
        o = __builtin_amdgcn_cvt_pk_u8_f32(lo.y, 1u, o);
        o = __builtin_amdgcn_cvt_pk_u8_f32(hi.x, 2u, o);
        return __builtin_amdgcn_cvt_pk_u8_f32(hi.y, 3u, o);
    }
    RF_DEV static void store(char* row, unsigned xoff, f4 v) { *reinterpret_cast<unsigned*>(row + xoff) = pack(v); }
    RF_DEV static void store_row(char* row, unsigned xoff, f4 v)       // wave-uniform row address: see PxF32::store_row
    {
        const unsigned d = pack(v);
        asm volatile(RF_SBASE "global_store_dword %[off], %[data], vcc" RF_STORE_MOD ::[off] "v"(xoff), [data] "v"(d), [base] "s"(row) : "memory", "vcc");
    }
    // what a store followed by a load of the next node does to a value: decode(pack(v)) without
    // the trip through the integer byte (the code is the same number either way)
    RF_DEV static f4 requant(f4 v)
    {
        const v2f lo = unorm8_code_to_f32_2(unorm8_code2(v2f{v.x, v.y}));
        const v2f hi = unorm8_code_to_f32_2(unorm8_code2(v2f{v.z, v.w}));
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
};

#ifndef __HIPCC_RTC__
// dense KxK convolution launch (rf_conv.hip), called from launch_ops
hipError_t launch_conv2d(int fmt, const Op& op, Image src, Image dst, const Geom& g, const StreamTuning& tune, hipStream_t stream);
#endif

}  // namespace rf
