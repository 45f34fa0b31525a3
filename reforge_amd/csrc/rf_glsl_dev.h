// rf_glsl_dev.h -- what a translated GLSL compute shader ({shader_path}/{type}.comp, rf_glsl.h) is compiled against: the GLSL
// types and built-in functions a filter uses, storage images, and the kernel that runs one invocation per thread.
//
// In the reference a filter type IS such a file: shaderc compiles it to SPIR-V (src/vulkan/shader.rs:73-93), spirv-reflect
// finds its bindings (shader.rs:106-160) and vkCmdDispatch runs ceil(W/16) x ceil(H/16) workgroups of it
// (src/vulkan/command.rs:166-194).  Here the translator (rf_glsl.cpp) rewrites the shader into a C++ struct template --
// `template <class RfgPx> struct RfgShader { <globals as members> <functions as member functions> void main(); }` inside
// namespace rfglsl, so that every GLSL name below is found before anything of the HIP headers -- and hiprtc compiles
// glsl_node_kernel<Px, Shader> at rf_graph_create (rf_jit.cpp), Px = the graph's texel format (rf_device.h: imageLoad /
// imageStore convert as DESIGN.md 3 says, whatever format qualifier the file carries -- shaders/passthrough.comp:4-5 says
// rgba8 and is run on rgba32f images by default, src/main.rs:60).
//
// Vectors are clang extended vectors: swizzles (.xyzw / .rgba, also as l-values), component-wise arithmetic and
// vector-scalar arithmetic come with the type.  Constructors are calls of mk_<type>(...), which the translator writes for
// every `vecN(...)` (a C++ cast between vectors would reinterpret bits).  -ffp-contract=off: a * b + c stays two
// roundings, fma() is one -- what `precise` asks for.
//
// The first half of this file (up to RFGLSL_KERNEL) has no HIP in it: the oracle compiles it for the host with clang++
// (oracle/glsl_host.py) to run the same translation on the CPU.
#pragma once

#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define RFG __device__ __forceinline__
#define RFGLSL_BUFFER_LOADS 1
#else
#include <math.h>
#include <string.h>
#define RFG inline
#endif

namespace rfglsl {

typedef unsigned int uint;
#define RFG_VEC(T, name)                                    \
    typedef T name##2 __attribute__((ext_vector_type(2))); \
    typedef T name##3 __attribute__((ext_vector_type(3))); \
    typedef T name##4 __attribute__((ext_vector_type(4)));
RFG_VEC(float, vec)
RFG_VEC(int, ivec)
RFG_VEC(unsigned, uvec)
typedef ivec2 bvec2;      // a boolean vector is an int vector holding 0 / 1
typedef ivec3 bvec3;
typedef ivec4 bvec4;
#undef RFG_VEC

// ---- constructors: the arguments' components in order, converted; one scalar fills every component ------------------------
template <class T> struct Comps {
    T v[16];
    int n = 0;
    RFG void put(float s) { v[n++] = (T)s; }
    RFG void put(double s) { v[n++] = (T)s; }
    RFG void put(int s) { v[n++] = (T)s; }
    RFG void put(unsigned s) { v[n++] = (T)s; }
    RFG void put(bool s) { v[n++] = (T)(s ? 1 : 0); }
#define RFG_PUT(V, N) \
    RFG void put(V s) { for (int i = 0; i < N; ++i) v[n++] = (T)s[i]; }
    RFG_PUT(vec2, 2) RFG_PUT(vec3, 3) RFG_PUT(vec4, 4) RFG_PUT(ivec2, 2) RFG_PUT(ivec3, 3) RFG_PUT(ivec4, 4) RFG_PUT(uvec2, 2) RFG_PUT(uvec3, 3) RFG_PUT(uvec4, 4)
#undef RFG_PUT
};
template <class V, class T, int N, class... A> RFG V mk_vector(A... a)
{
    static_assert(sizeof...(A) >= 1, "a constructor needs an argument");
    Comps<T> c;
    (c.put(a), ...);
    V r;
    for (int i = 0; i < N; ++i) r[i] = c.n == 1 ? c.v[0] : c.v[i];
    return r;
}
#define RFG_MK(V, T, N) \
    template <class... A> RFG V mk_##V(A... a) { return mk_vector<V, T, N>(a...); }
RFG_MK(vec2, float, 2) RFG_MK(vec3, float, 3) RFG_MK(vec4, float, 4)
RFG_MK(ivec2, int, 2) RFG_MK(ivec3, int, 3) RFG_MK(ivec4, int, 4)
RFG_MK(uvec2, unsigned, 2) RFG_MK(uvec3, unsigned, 3) RFG_MK(uvec4, unsigned, 4)
#undef RFG_MK
template <class V, int N, class... A> RFG V mk_bvector(A... a)
{
    Comps<float> c;
    (c.put(a), ...);
    V r;
    for (int i = 0; i < N; ++i) r[i] = (c.n == 1 ? c.v[0] : c.v[i]) != 0.0f ? 1 : 0;
    return r;
}
template <class... A> RFG bvec2 mk_bvec2(A... a) { return mk_bvector<bvec2, 2>(a...); }
template <class... A> RFG bvec3 mk_bvec3(A... a) { return mk_bvector<bvec3, 3>(a...); }
template <class... A> RFG bvec4 mk_bvec4(A... a) { return mk_bvector<bvec4, 4>(a...); }

// ---- component-wise built-ins ----------------------------------------------------------------------------------------------
#define RFG_MAP1(V, N, name) \
    RFG V name(V a) { V r; for (int i = 0; i < N; ++i) r[i] = name(a[i]); return r; }
#define RFG_MAP2(V, N, name) \
    RFG V name(V a, V b) { V r; for (int i = 0; i < N; ++i) r[i] = name(a[i], b[i]); return r; }
#define RFG_MAP2S(V, N, S, name) \
    RFG V name(V a, S b) { V r; for (int i = 0; i < N; ++i) r[i] = name(a[i], b); return r; }
#define RFG_MAP3(V, N, name) \
    RFG V name(V a, V b, V c) { V r; for (int i = 0; i < N; ++i) r[i] = name(a[i], b[i], c[i]); return r; }
#define RFG_MAP3S(V, N, S, name) \
    RFG V name(V a, S b, S c) { V r; for (int i = 0; i < N; ++i) r[i] = name(a[i], b, c); return r; }
#define RFG_V1(P, name) RFG_MAP1(P##2, 2, name) RFG_MAP1(P##3, 3, name) RFG_MAP1(P##4, 4, name)
#define RFG_V2(P, name) RFG_MAP2(P##2, 2, name) RFG_MAP2(P##3, 3, name) RFG_MAP2(P##4, 4, name)
#define RFG_V2S(P, S, name) RFG_MAP2S(P##2, 2, S, name) RFG_MAP2S(P##3, 3, S, name) RFG_MAP2S(P##4, 4, S, name)
#define RFG_V3(P, name) RFG_MAP3(P##2, 2, name) RFG_MAP3(P##3, 3, name) RFG_MAP3(P##4, 4, name)
#define RFG_V3S(P, S, name) RFG_MAP3S(P##2, 2, S, name) RFG_MAP3S(P##3, 3, S, name) RFG_MAP3S(P##4, 4, S, name)
#define RFG_FV1(name) RFG_V1(vec, name)
#define RFG_FV2(name) RFG_V2(vec, name)
#define RFG_FV2S(name) RFG_V2S(vec, float, name)
#define RFG_FV3(name) RFG_V3(vec, name)
#define RFG_FV3S(name) RFG_V3S(vec, float, name)

RFG float abs(float x) { return ::fabsf(x); }
RFG int abs(int x) { return x < 0 ? -x : x; }
RFG float sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
RFG int sign(int x) { return x > 0 ? 1 : (x < 0 ? -1 : 0); }
RFG float floor(float x) { return ::floorf(x); }
RFG float ceil(float x) { return ::ceilf(x); }
RFG float trunc(float x) { return ::truncf(x); }
RFG float roundEven(float x) { return ::rintf(x); }
RFG float round(float x) { return ::rintf(x); }          // "the fraction 0.5 rounds in a direction chosen by the implementation"
RFG float fract(float x) { return x - ::floorf(x); }
RFG float sqrt(float x) { return ::sqrtf(x); }
RFG float inversesqrt(float x) { return 1.0f / ::sqrtf(x); }
RFG float exp(float x) { return ::expf(x); }
RFG float exp2(float x) { return ::exp2f(x); }
RFG float log(float x) { return ::logf(x); }
RFG float log2(float x) { return ::log2f(x); }
RFG float sin(float x) { return ::sinf(x); }
RFG float cos(float x) { return ::cosf(x); }
RFG float tan(float x) { return ::tanf(x); }
RFG float asin(float x) { return ::asinf(x); }
RFG float acos(float x) { return ::acosf(x); }
RFG float atan(float x) { return ::atanf(x); }
RFG float sinh(float x) { return ::sinhf(x); }
RFG float cosh(float x) { return ::coshf(x); }
RFG float tanh(float x) { return ::tanhf(x); }
RFG float radians(float x) { return x * 0.017453292519943295f; }
RFG float degrees(float x) { return x * 57.29577951308232f; }
RFG_FV1(abs) RFG_FV1(sign) RFG_FV1(floor) RFG_FV1(ceil) RFG_FV1(trunc) RFG_FV1(roundEven) RFG_FV1(round) RFG_FV1(fract) RFG_FV1(sqrt)
RFG_FV1(inversesqrt) RFG_FV1(exp) RFG_FV1(exp2) RFG_FV1(log) RFG_FV1(log2) RFG_FV1(sin) RFG_FV1(cos) RFG_FV1(tan) RFG_FV1(asin) RFG_FV1(acos)
RFG_FV1(atan) RFG_FV1(sinh) RFG_FV1(cosh) RFG_FV1(tanh) RFG_FV1(radians) RFG_FV1(degrees)
RFG_V1(ivec, abs) RFG_V1(ivec, sign)

// min / max: "y if y < x, otherwise x" / "y if x < y, otherwise x" (the GLSL wording; it decides what a NaN operand gives)
RFG float min(float x, float y) { return y < x ? y : x; }
RFG float max(float x, float y) { return x < y ? y : x; }
RFG int min(int x, int y) { return y < x ? y : x; }
RFG int max(int x, int y) { return x < y ? y : x; }
RFG uint min(uint x, uint y) { return y < x ? y : x; }
RFG uint max(uint x, uint y) { return x < y ? y : x; }
RFG float mod(float x, float y) { return x - y * ::floorf(x / y); }
RFG float pow(float x, float y) { return ::powf(x, y); }
RFG float atan(float y, float x) { return ::atan2f(y, x); }
RFG float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
RFG_FV2(min) RFG_FV2(max) RFG_FV2(mod) RFG_FV2(pow) RFG_FV2(atan) RFG_FV2(step) RFG_FV2S(min) RFG_FV2S(max) RFG_FV2S(mod)
RFG_V2(ivec, min) RFG_V2(ivec, max) RFG_V2S(ivec, int, min) RFG_V2S(ivec, int, max) RFG_V2(uvec, min) RFG_V2(uvec, max) RFG_V2S(uvec, uint, min) RFG_V2S(uvec, uint, max)
RFG vec2 step(float e, vec2 x) { return step(mk_vec2(e), x); }
RFG vec3 step(float e, vec3 x) { return step(mk_vec3(e), x); }
RFG vec4 step(float e, vec4 x) { return step(mk_vec4(e), x); }

RFG float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
RFG int clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
RFG uint clamp(uint x, uint lo, uint hi) { return min(max(x, lo), hi); }
// min / max / clamp of scalars of DIFFERENT types (`max(x, 0)`, `clamp(i, 0.0, 1.0)`): GLSL converts int -> uint -> float and takes the
// overload of the widest operand; C++ would call the three overloads above ambiguous
template <class T> struct rfg_rank { static constexpr int v = -1; };
template <> struct rfg_rank<int> { static constexpr int v = 0; };
template <> struct rfg_rank<uint> { static constexpr int v = 1; };
template <> struct rfg_rank<float> { static constexpr int v = 2; };
template <int R> struct rfg_of_rank {};
template <> struct rfg_of_rank<0> { typedef int type; };
template <> struct rfg_of_rank<1> { typedef uint type; };
template <> struct rfg_of_rank<2> { typedef float type; };
template <class A, class B, class C = A> struct rfg_widest {
    static constexpr int a = rfg_rank<A>::v, b = rfg_rank<B>::v, c = rfg_rank<C>::v;
    static constexpr bool mixed = a >= 0 && b >= 0 && c >= 0 && !(a == b && b == c);
    static constexpr int r = a > b ? (a > c ? a : c) : (b > c ? b : c);
};
template <bool M, int R> struct rfg_mixed {};
template <int R> struct rfg_mixed<true, R> { typedef typename rfg_of_rank<R>::type type; };
template <class A, class B> RFG typename rfg_mixed<rfg_widest<A, B>::mixed, rfg_widest<A, B>::r>::type min(A x, B y)
{
    typedef typename rfg_of_rank<rfg_widest<A, B>::r>::type T;
    return min((T)x, (T)y);
}
template <class A, class B> RFG typename rfg_mixed<rfg_widest<A, B>::mixed, rfg_widest<A, B>::r>::type max(A x, B y)
{
    typedef typename rfg_of_rank<rfg_widest<A, B>::r>::type T;
    return max((T)x, (T)y);
}
template <class A, class B, class C> RFG typename rfg_mixed<rfg_widest<A, B, C>::mixed, rfg_widest<A, B, C>::r>::type clamp(A x, B lo, C hi)
{
    typedef typename rfg_of_rank<rfg_widest<A, B, C>::r>::type T;
    return clamp((T)x, (T)lo, (T)hi);
}
RFG float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RFG float mix(float x, float y, bool a) { return a ? y : x; }
RFG float smoothstep(float e0, float e1, float x)
{
    const float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
#ifndef RFG_SPLIT_FMA
RFG float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
RFG vec2 fma(vec2 a, vec2 b, vec2 c) { return __builtin_elementwise_fma(a, b, c); }
RFG vec3 fma(vec3 a, vec3 b, vec3 c) { return __builtin_elementwise_fma(a, b, c); }
RFG vec4 fma(vec4 a, vec4 b, vec4 c) { return __builtin_elementwise_fma(a, b, c); }
#else   // tests only (tests/test_glsl_mesa.py, host compile): fma() as a software rasteriser without fused hardware evaluates it -- two roundings
RFG float fma(float a, float b, float c) { return a * b + c; }
RFG vec2 fma(vec2 a, vec2 b, vec2 c) { return a * b + c; }
RFG vec3 fma(vec3 a, vec3 b, vec3 c) { return a * b + c; }
RFG vec4 fma(vec4 a, vec4 b, vec4 c) { return a * b + c; }
#endif
RFG_FV3(clamp) RFG_FV3S(clamp) RFG_FV3(mix) RFG_FV3(smoothstep)
RFG_V3(ivec, clamp) RFG_V3S(ivec, int, clamp) RFG_V3(uvec, clamp) RFG_V3S(uvec, uint, clamp)
RFG vec2 mix(vec2 x, vec2 y, float a) { return mix(x, y, mk_vec2(a)); }
RFG vec3 mix(vec3 x, vec3 y, float a) { return mix(x, y, mk_vec3(a)); }
RFG vec4 mix(vec4 x, vec4 y, float a) { return mix(x, y, mk_vec4(a)); }
RFG vec2 mix(vec2 x, vec2 y, bvec2 a) { vec2 r; for (int i = 0; i < 2; ++i) r[i] = a[i] ? y[i] : x[i]; return r; }
RFG vec3 mix(vec3 x, vec3 y, bvec3 a) { vec3 r; for (int i = 0; i < 3; ++i) r[i] = a[i] ? y[i] : x[i]; return r; }
RFG vec4 mix(vec4 x, vec4 y, bvec4 a) { vec4 r; for (int i = 0; i < 4; ++i) r[i] = a[i] ? y[i] : x[i]; return r; }
RFG vec2 smoothstep(float a, float b, vec2 x) { return smoothstep(mk_vec2(a), mk_vec2(b), x); }
RFG vec3 smoothstep(float a, float b, vec3 x) { return smoothstep(mk_vec3(a), mk_vec3(b), x); }
RFG vec4 smoothstep(float a, float b, vec4 x) { return smoothstep(mk_vec4(a), mk_vec4(b), x); }

RFG bool isnan(float x) { return x != x; }
RFG bool isinf(float x) { return ::fabsf(x) == __builtin_huge_valf(); }
RFG int floatBitsToInt(float x) { int r; __builtin_memcpy(&r, &x, 4); return r; }
RFG uint floatBitsToUint(float x) { uint r; __builtin_memcpy(&r, &x, 4); return r; }
RFG float intBitsToFloat(int x) { float r; __builtin_memcpy(&r, &x, 4); return r; }
RFG float uintBitsToFloat(uint x) { float r; __builtin_memcpy(&r, &x, 4); return r; }
#define RFG_BITS(name, VI, VO) \
    RFG VO##2 name(VI##2 a) { VO##2 r; __builtin_memcpy(&r, &a, sizeof(r)); return r; } \
    RFG VO##3 name(VI##3 a) { VO##3 r; for (int i = 0; i < 3; ++i) r[i] = name(a[i]); return r; } \
    RFG VO##4 name(VI##4 a) { VO##4 r; __builtin_memcpy(&r, &a, sizeof(r)); return r; }
RFG_BITS(floatBitsToInt, vec, ivec) RFG_BITS(floatBitsToUint, vec, uvec) RFG_BITS(intBitsToFloat, ivec, vec) RFG_BITS(uintBitsToFloat, uvec, vec)
#undef RFG_BITS

// ---- geometric --------------------------------------------------------------------------------------------------------------
RFG float dot(float a, float b) { return a * b; }
RFG float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
RFG float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RFG float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
RFG float length(float a) { return ::fabsf(a); }
RFG float length(vec2 a) { return ::sqrtf(dot(a, a)); }
RFG float length(vec3 a) { return ::sqrtf(dot(a, a)); }
RFG float length(vec4 a) { return ::sqrtf(dot(a, a)); }
RFG float distance(float a, float b) { return length(a - b); }
RFG float distance(vec2 a, vec2 b) { return length(a - b); }
RFG float distance(vec3 a, vec3 b) { return length(a - b); }
RFG float distance(vec4 a, vec4 b) { return length(a - b); }
RFG vec2 normalize(vec2 a) { return a / length(a); }
RFG vec3 normalize(vec3 a) { return a / length(a); }
RFG vec4 normalize(vec4 a) { return a / length(a); }
RFG vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RFG vec2 reflect(vec2 i, vec2 n) { return i - 2.0f * dot(n, i) * n; }
RFG vec3 reflect(vec3 i, vec3 n) { return i - 2.0f * dot(n, i) * n; }
RFG vec4 reflect(vec4 i, vec4 n) { return i - 2.0f * dot(n, i) * n; }

// ---- relational -------------------------------------------------------------------------------------------------------------
#define RFG_REL(name, op, V, B, N) \
    RFG B name(V a, V b) { B r; for (int i = 0; i < N; ++i) r[i] = a[i] op b[i] ? 1 : 0; return r; }
#define RFG_RELS(name, op)                                                                                                     \
    RFG_REL(name, op, vec2, bvec2, 2) RFG_REL(name, op, vec3, bvec3, 3) RFG_REL(name, op, vec4, bvec4, 4)                      \
    RFG_REL(name, op, ivec2, bvec2, 2) RFG_REL(name, op, ivec3, bvec3, 3) RFG_REL(name, op, ivec4, bvec4, 4)                   \
    RFG_REL(name, op, uvec2, bvec2, 2) RFG_REL(name, op, uvec3, bvec3, 3) RFG_REL(name, op, uvec4, bvec4, 4)
RFG_RELS(lessThan, <) RFG_RELS(lessThanEqual, <=) RFG_RELS(greaterThan, >) RFG_RELS(greaterThanEqual, >=) RFG_RELS(equal, ==) RFG_RELS(notEqual, !=)
#undef RFG_RELS
#undef RFG_REL
RFG bool any(bvec2 a) { return (a.x | a.y) != 0; }
RFG bool any(bvec3 a) { return (a.x | a.y | a.z) != 0; }
RFG bool any(bvec4 a) { return (a.x | a.y | a.z | a.w) != 0; }
RFG bool all(bvec2 a) { return a.x != 0 && a.y != 0; }
RFG bool all(bvec3 a) { return a.x != 0 && a.y != 0 && a.z != 0; }
RFG bool all(bvec4 a) { return a.x != 0 && a.y != 0 && a.z != 0 && a.w != 0; }
RFG bvec2 rfg_not(bvec2 a) { return bvec2{a.x == 0, a.y == 0}; }       // GLSL's not(): `not` is an operator spelling in C++, the translator renames the call
RFG bvec3 rfg_not(bvec3 a) { return bvec3{a.x == 0, a.y == 0, a.z == 0}; }
RFG bvec4 rfg_not(bvec4 a) { return bvec4{a.x == 0, a.y == 0, a.z == 0, a.w == 0}; }
// Atomic memory functions (GLSL 4.50 8.11) on the int / uint members of storage blocks and on shared variables; the translator renames the
// calls (HIP's functions of the same names take pointers).  Each returns the value the memory held before.  On the host (tests) the
// invocations run one after another, so the plain operation is the atomic one.
#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define RFG_ATOMIC(name, hipname, expr) \
    template <class T, class V> RFG T rfg_##name(T& m, V v) { static_assert(sizeof(T) == 4 && (T)0.5 == (T)0, "atomic functions take int or uint memory"); return ::hipname(&m, (T)v); }
#else
#define RFG_ATOMIC(name, hipname, expr) \
    template <class T, class V> RFG T rfg_##name(T& m, V v) { static_assert(sizeof(T) == 4 && (T)0.5 == (T)0, "atomic functions take int or uint memory"); const T o = m, d = (T)v; m = (expr); return o; }
#endif
RFG_ATOMIC(atomicAdd, atomicAdd, (T)(o + d)) RFG_ATOMIC(atomicMin, atomicMin, d < o ? d : o) RFG_ATOMIC(atomicMax, atomicMax, d > o ? d : o)
RFG_ATOMIC(atomicAnd, atomicAnd, (T)(o & d)) RFG_ATOMIC(atomicOr, atomicOr, (T)(o | d)) RFG_ATOMIC(atomicXor, atomicXor, (T)(o ^ d))
RFG_ATOMIC(atomicExchange, atomicExch, d)
#undef RFG_ATOMIC
template <class T, class C, class V> RFG T rfg_atomicCompSwap(T& m, C compare, V v)
{
    static_assert(sizeof(T) == 4 && (T)0.5 == (T)0, "atomic functions take int or uint memory");
#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
    return ::atomicCAS(&m, (T)compare, (T)v);
#else
    const T o = m;
    if (o == (T)compare) m = (T)v;
    return o;
#endif
}
RFG bvec2 isnan(vec2 a) { return bvec2{a.x != a.x, a.y != a.y}; }
RFG bvec3 isnan(vec3 a) { return bvec3{a.x != a.x, a.y != a.y, a.z != a.z}; }
RFG bvec4 isnan(vec4 a) { return bvec4{a.x != a.x, a.y != a.y, a.z != a.z, a.w != a.w}; }

// ---- matrices: column vectors, m[c][r] as in GLSL ------------------------------------------------------------------------------
template <class V, int N> struct matN {
    V c[N];
    RFG V& operator[](int i) { return c[i]; }
    RFG const V& operator[](int i) const { return c[i]; }
};
typedef matN<vec2, 2> mat2;
typedef matN<vec3, 3> mat3;
typedef matN<vec4, 4> mat4;
template <class M, class V, int N, class... A> RFG M mk_matrix(A... a)
{
    Comps<float> c;
    (c.put(a), ...);
    M m;
    for (int col = 0; col < N; ++col)
        for (int row = 0; row < N; ++row) m.c[col][row] = c.n == 1 ? (col == row ? c.v[0] : 0.0f) : c.v[col * N + row];
    return m;
}
template <class... A> RFG mat2 mk_mat2(A... a) { return mk_matrix<mat2, vec2, 2>(a...); }
template <class... A> RFG mat3 mk_mat3(A... a) { return mk_matrix<mat3, vec3, 3>(a...); }
template <class... A> RFG mat4 mk_mat4(A... a) { return mk_matrix<mat4, vec4, 4>(a...); }
RFG mat3 mk_mat3(mat4 m) { mat3 r; for (int i = 0; i < 3; ++i) r.c[i] = m.c[i].xyz; return r; }
RFG mat2 mk_mat2(mat3 m) { mat2 r; for (int i = 0; i < 2; ++i) r.c[i] = m.c[i].xy; return r; }
template <class V, int N> RFG V operator*(const matN<V, N>& m, V v)      // columns scaled and summed left to right
{
    V r = m.c[0] * v[0];
    for (int i = 1; i < N; ++i) r = r + m.c[i] * v[i];
    return r;
}
template <class V, int N> RFG V operator*(V v, const matN<V, N>& m)
{
    V r;
    for (int i = 0; i < N; ++i) r[i] = dot(v, m.c[i]);
    return r;
}
template <class V, int N> RFG matN<V, N> operator*(const matN<V, N>& a, const matN<V, N>& b)
{
    matN<V, N> r;
    for (int i = 0; i < N; ++i) r.c[i] = a * b.c[i];
    return r;
}
template <class V, int N> RFG matN<V, N> operator*(const matN<V, N>& a, float s) { matN<V, N> r; for (int i = 0; i < N; ++i) r.c[i] = a.c[i] * s; return r; }
template <class V, int N> RFG matN<V, N> operator*(float s, const matN<V, N>& a) { return a * s; }
template <class V, int N> RFG matN<V, N> operator+(const matN<V, N>& a, const matN<V, N>& b) { matN<V, N> r; for (int i = 0; i < N; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
template <class V, int N> RFG matN<V, N> operator-(const matN<V, N>& a, const matN<V, N>& b) { matN<V, N> r; for (int i = 0; i < N; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
template <class V, int N> RFG matN<V, N> transpose(const matN<V, N>& a)
{
    matN<V, N> r;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) r.c[i][j] = a.c[j][i];
    return r;
}

// ---- == and != (the translator writes every one as a call): one bool, also for vectors and matrices --------------------------------------
template <class A, class B> RFG bool rfg_eq(A a, B b) { return a == b; }
#define RFG_EQ(V, N) \
    RFG bool rfg_eq(V a, V b) { bool r = true; for (int i = 0; i < N; ++i) r = r && a[i] == b[i]; return r; }
RFG_EQ(vec2, 2) RFG_EQ(vec3, 3) RFG_EQ(vec4, 4) RFG_EQ(ivec2, 2) RFG_EQ(ivec3, 3) RFG_EQ(ivec4, 4) RFG_EQ(uvec2, 2) RFG_EQ(uvec3, 3) RFG_EQ(uvec4, 4)
#undef RFG_EQ
template <class V, int N> RFG bool rfg_eq(const matN<V, N>& a, const matN<V, N>& b) { bool r = true; for (int i = 0; i < N; ++i) r = r && rfg_eq(a.c[i], b.c[i]); return r; }
// An array with a stated size: a value, as in GLSL (copied when assigned, passed or returned; compared element by element).  An aggregate:
// `rfg_arr<float, 3> w = {a, b, c}` and `return rfg_arr<float, 2>{x, y}` are what the translator writes for GLSL's array constructors.
template <class T, int N> struct rfg_arr {
    T v[N];
    template <class I> RFG T& operator[](I i) { return v[i]; }
    template <class I> RFG const T& operator[](I i) const { return v[i]; }
};
template <class T, int N> RFG bool rfg_eq(const rfg_arr<T, N>& a, const rfg_arr<T, N>& b)
{
    bool r = true;
    for (int i = 0; i < N; ++i) r = r && rfg_eq(a.v[i], b.v[i]);
    return r;
}
template <class T, int N> RFG bool rfg_eq(const T (&a)[N], const T (&b)[N])      // arrays compare element by element (C++ would compare their addresses)
{
    bool r = true;
    for (int i = 0; i < N; ++i) r = r && rfg_eq(a[i], b[i]);
    return r;
}
template <class A, class B> RFG bool rfg_ne(const A& a, const B& b) { return !rfg_eq(a, b); }
RFG bool rfg_xor(bool a, bool b) { return a != b; }      // a ^^ b

// name.length() (the translator writes rfg_length(name)): elements of an array, components of a vector
template <class T, int N> RFG constexpr int rfg_length(const T (&)[N]) { return N; }
template <class T, int N> RFG constexpr int rfg_length(const rfg_arr<T, N>&) { return N; }
RFG constexpr int rfg_length(vec2) { return 2; }
RFG constexpr int rfg_length(vec3) { return 3; }
RFG constexpr int rfg_length(vec4) { return 4; }
RFG constexpr int rfg_length(ivec2) { return 2; }
RFG constexpr int rfg_length(ivec3) { return 3; }
RFG constexpr int rfg_length(ivec4) { return 4; }
RFG constexpr int rfg_length(uvec2) { return 2; }
RFG constexpr int rfg_length(uvec3) { return 3; }
RFG constexpr int rfg_length(uvec4) { return 4; }

// ---- integer and packing built-ins --------------------------------------------------------------------------------------------------------
RFG int bitCount(uint x) { return __builtin_popcount(x); }
RFG int bitCount(int x) { return __builtin_popcount((unsigned)x); }
RFG int findLSB(uint x) { return x == 0u ? -1 : __builtin_ctz(x); }
RFG int findLSB(int x) { return findLSB((uint)x); }
RFG int findMSB(uint x) { return x == 0u ? -1 : 31 - __builtin_clz(x); }
RFG int findMSB(int x) { return x < 0 ? findMSB(~(uint)x) : findMSB((uint)x); }
RFG uint bitfieldExtract(uint v, int offset, int bits) { return bits == 0 ? 0u : (v >> offset) & (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u)); }
RFG int bitfieldExtract(int v, int offset, int bits) { return bits == 0 ? 0 : (int)((uint)v << (32 - offset - bits)) >> (32 - bits); }
RFG uint bitfieldInsert(uint base, uint ins, int offset, int bits)
{
    const uint m = bits >= 32 ? 0xffffffffu : (((1u << bits) - 1u) << offset);
    return bits == 0 ? base : (base & ~m) | ((ins << offset) & m);
}
RFG uint packUnorm4x8(vec4 v)
{
    uint r = 0u;
    for (int i = 0; i < 4; ++i) r |= (uint)::rintf(clamp(v[i], 0.0f, 1.0f) * 255.0f) << (8 * i);
    return r;
}
RFG vec4 unpackUnorm4x8(uint p) { return vec4{(float)(p & 255u), (float)((p >> 8) & 255u), (float)((p >> 16) & 255u), (float)(p >> 24)} / 255.0f; }
RFG float determinant(const mat2& m) { return m.c[0].x * m.c[1].y - m.c[1].x * m.c[0].y; }
RFG float determinant(const mat3& m) { return dot(m.c[0], cross(m.c[1], m.c[2])); }
RFG mat2 inverse(const mat2& m)
{
    const float d = determinant(m);
    mat2 r;
    r.c[0] = vec2{m.c[1].y, -m.c[0].y} / d;
    r.c[1] = vec2{-m.c[1].x, m.c[0].x} / d;
    return r;
}
RFG mat3 inverse(const mat3& m)
{
    const float d = determinant(m);
    mat3 t;      // rows of the inverse = cross products of the columns, over the determinant
    t.c[0] = cross(m.c[1], m.c[2]) / d;
    t.c[1] = cross(m.c[2], m.c[0]) / d;
    t.c[2] = cross(m.c[0], m.c[1]) / d;
    return transpose(t);
}

// ---- storage images -------------------------------------------------------------------------------------------------------------
// `uniform image2D name`: the allocated image the graph wires to the variable's binding.  Coordinates are FRAME coordinates
// (imageSize = the whole frame); a rank of a row-strip partition holds rows [y_first - ghost, y_last + ghost] of it
// (row_lo .. row_hi), which `#pragma rf radius N` promises to be enough.  A load outside the image returns zero, a store
// outside it is dropped (what Vulkan's robust image access gives; the reference leaves it to the driver).
template <class Px> struct image2D {
    char* base;                   // address of frame row 0 (rows outside [row_lo, row_hi] are not this rank's to touch)
    unsigned long long pitch;
    int W, H, row_lo, row_hi;     // frame size; frame rows this rank may read
    int wr_lo, wr_hi;             // frame rows this LAUNCH writes, inclusive (the whole frame on one GPU; a strip's rows, or one part of a split launch)
    const char* zero;             // 16 zero bytes: what a load outside the image reads
#ifdef RFGLSL_BUFFER_LOADS
    // the image as a BUFFER RESOURCE: base = frame row 0, range = up to the last readable row.  A load is then one 32-bit byte
    // offset (no 64-bit address arithmetic), and an offset outside the range -- which is where a load outside the image is sent --
    // returns zero by the hardware's own range check: what Vulkan's robust access gives.  Images below 4 GiB only (Px::BUFFER).
    __amdgpu_buffer_rsrc_t rsrc;
    RFG void finish()
    {
        if constexpr (Px::BUFFER) rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)((unsigned)(row_hi + 1) * (unsigned)pitch), 0x00020000);
    }
#else
    RFG void finish() {}
#endif
};
template <class Px> RFG ivec2 imageSize(const image2D<Px>& im) { return ivec2{im.W, im.H}; }
// BRANCH-FREE: the coordinates are clamped into what may be read, the texel is loaded whatever they were, and a load outside
// the image selects zero afterwards -- a filter calls this 9, 25, 81 times per invocation, and a load under a branch of its own
// is waited for before the next one is issued (the compiler cannot move a load across the branch that guards it).
template <class Px> RFG vec4 imageLoad(const image2D<Px>& im, ivec2 p)
{
    // (a readable image is always wired: the planner refuses a graph that leaves one open, rf_plan.cpp -- no test of `base` here: a
    // branch per load, even a uniform one, makes the loads of a filter wait for one another)
    const bool in_frame = (unsigned)p.x < (unsigned)im.W && (unsigned)(p.y - im.row_lo) <= (unsigned)(im.row_hi - im.row_lo);
    // rows are frame rows (>= 0 wherever they may be read) and a pitch is below 4 GiB: one 32 x 32 -> 64 bit multiply-add.  A load
    // outside the image reads the zero texel the graph keeps for that (one select on the address; no clamps, no select on the result)
#ifdef RFGLSL_BUFFER_LOADS
    if constexpr (Px::BUFFER) {
        const unsigned off = in_frame ? (unsigned)p.y * (unsigned)im.pitch + (unsigned)p.x * (unsigned)Px::BPP : 0xffffffffu;
        typename Px::Raw raw;
        if constexpr (Px::BPP == 16) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(im.rsrc, (int)off, 0, 0);
            __builtin_memcpy(&raw, &v, 16);
        } else {
            const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(im.rsrc, (int)off, 0, 0);
            __builtin_memcpy(&raw, &v, 4);
        }
        const auto t = Px::decode(raw);
        return vec4{t.x, t.y, t.z, t.w};
    }
#endif
    const char* at = im.base + ((unsigned long long)(unsigned)p.y * (unsigned long long)(unsigned)im.pitch + (unsigned long long)((unsigned)p.x * (unsigned)Px::BPP));
    const auto t = Px::decode(Px::load(in_frame ? at : im.zero, 0u));
    return vec4{t.x, t.y, t.z, t.w};
}
template <class Px> RFG void imageStore(const image2D<Px>& im, ivec2 p, vec4 v)
{
    if ((unsigned)p.x >= (unsigned)im.W || (unsigned)p.y >= (unsigned)im.H || p.y < im.wr_lo || p.y > im.wr_hi || !im.base) return;
    Px::store(im.base + (long long)p.y * (long long)im.pitch, (unsigned)p.x * (unsigned)Px::BPP, Px::texel(v.x, v.y, v.z, v.w));
}

// `uniform sampler2D name`: the same allocated image bound as a COMBINED IMAGE SAMPLER (shader.rs:98; pipeline_graph.rs:98-105 writes
// every image descriptor with the graph's one sampler).  That sampler is LINEAR / LINEAR, address mode U = CLAMP_TO_EDGE, and V = the
// Vulkan default REPEAT: vkutils.rs:358-365 sets `address_mode_u` once and `address_mode_w` twice, never `address_mode_v` -- restated as
// it is.  texture(): unnormalised coordinate = uv * size - 0.5, the four texels around it weighted (1-a)(1-b), a(1-b), (1-a)b, ab
// (Vulkan 1.3, 16.8 "Texel filtering") in float; hardware filters with 8-bit fixed-point weights, so the low bits of a filtered sample
// are the implementation's there (parity unpinned).  texelFetch(): the texel itself; outside the image: zero.
template <class Px> struct sampler2D {
    image2D<Px> im;
};
template <class Px> RFG ivec2 textureSize(const sampler2D<Px>& s, int) { return ivec2{s.im.W, s.im.H}; }
template <class Px> RFG vec4 texelFetch(const sampler2D<Px>& s, ivec2 p, int) { return imageLoad(s.im, p); }
template <class Px> RFG vec4 texture(const sampler2D<Px>& s, vec2 uv)
{
    const float u = uv.x * (float)s.im.W - 0.5f, v = uv.y * (float)s.im.H - 0.5f;
    const float fu = ::floorf(u), fv = ::floorf(v);
    const float a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > s.im.W - 1 ? s.im.W - 1 : i0);      // U: clamp to edge
    i1 = i1 < 0 ? 0 : (i1 > s.im.W - 1 ? s.im.W - 1 : i1);
    j0 %= s.im.H; j0 += j0 < 0 ? s.im.H : 0;                    // V: repeat
    j1 %= s.im.H; j1 += j1 < 0 ? s.im.H : 0;
    const vec4 t00 = imageLoad(s.im, ivec2{i0, j0}), t10 = imageLoad(s.im, ivec2{i1, j0}), t01 = imageLoad(s.im, ivec2{i0, j1}), t11 = imageLoad(s.im, ivec2{i1, j1});
    return ((1.0f - a) * (1.0f - b)) * t00 + (a * (1.0f - b)) * t10 + ((1.0f - a) * b) * t01 + (a * b) * t11;
}
template <class Px> RFG vec4 textureLod(const sampler2D<Px>& s, vec2 uv, float) { return texture(s, uv); }      // one level

// A shader the translator recognises as a POINT operation (rf_glsl.cpp, point_shader) also runs as a row stage of the stream kernel:
// its image variables then hold ONE texel -- the one the stage is handed, the one it hands on -- and the frame guard never fires.
struct PointPx {};
template <> struct image2D<PointPx> {
    mutable vec4 value;
};
RFG ivec2 imageSize(const image2D<PointPx>&) { return ivec2{0x7fffffff, 0x7fffffff}; }
RFG vec4 imageLoad(const image2D<PointPx>& im, ivec2) { return im.value; }
RFG void imageStore(const image2D<PointPx>& im, ivec2, vec4 v) { im.value = v; }

// what the kernel hands a shader object (GlslArgs below, the oracle's driver on the host)
struct GlslImage {
    char* base;
    unsigned long long pitch;
};
struct GlslFrame {
    int W, H;                 // the frame (imageSize)
    int row_lo, row_hi;       // frame rows this rank may read
    int y0, y1;               // rows whose invocations this launch runs (and may write); the launch that holds the frame's last row also runs
                              // the rows below it that the dispatch covers (y1 = groups_y * local_size_y): they exist in the reference too and
                              // may write storage blocks
    int groups_x, groups_y;   // the dispatch: ceil(W/16) x ceil(H/16) workgroups (command.rs:167-168) whatever local_size says
    const char* zero;         // 16 zero bytes in device memory (what imageLoad outside an image reads)
    int ring;                 // > 0: the launch runs only the invocations within `ring` texels of the frame's edges (rows y0 .. y1): the border
                              // of a node whose interior the window kernel computes (WinPx below)
    int pad;
};

}  // namespace rfglsl

// =====================================================================================================================================
#ifdef __HIPCC_RTC__
#define RFGLSL_KERNEL 1
namespace rfglsl {

// Texel formats with the constructor imageStore needs (rf_device.h)
template <class P, bool BUF> struct GPx : P {      // the texel formats of rf_device.h with the constructor imageStore needs
    static constexpr bool BUFFER = BUF;            // imageLoad through buffer resources (images below 4 GiB)
    RFG static rf::f4 texel(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
};

// A shader the translator recognises as a TRANSLATION-INVARIANT STENCIL (rf_glsl.cpp, stencil_shader) also runs on the LDS-tiled window
// kernel of the stage files (rf::user_node_kernel, rf_user_dev.h): there an invocation lives in a virtual frame of (2R+1)^2 texels around
// itself -- gl_GlobalInvocationID = (R, R), imageSize = (2R+1, 2R+1) -- so every coordinate it computes is a constant of the unrolled code
// and imageLoad is Window::at at a constant offset: a register of the window the thread slides down its outputs (R <= 2), or an LDS
// read at an immediate offset.  (Offsets are clamped into the window: a shader that reads further than it states stays inside memory;
// rf_graph_create compares this kernel with the generic one on a random frame and keeps it only if they agree.)
// ... and a recognised stencil of RADIUS 1 with one image in and one out is also a 3 x 3 ROW STAGE of the stream kernel (StUser, R = 1): the
// same virtual frame, 3 x 3, backed by the neighbourhood the stage is handed (clamp-to-edge at the frame's edges, as every stencil of the
// stream kernel sees it -- a file that treats the edges otherwise is found out by rf_graph_create and keeps a kernel of its own).
struct BoxPx {};
template <> struct image2D<BoxPx> {
    const rf::f4 (*n)[3];
    mutable vec4 value;
};
RFG ivec2 imageSize(const image2D<BoxPx>&) { return ivec2{3, 3}; }
RFG vec4 imageLoad(const image2D<BoxPx>& im, ivec2 q)
{
    const bool in = (unsigned)q.x < 3u && (unsigned)q.y < 3u;
    const int x = q.x < 0 ? 0 : (q.x > 2 ? 2 : q.x), y = q.y < 0 ? 0 : (q.y > 2 ? 2 : q.y);
    const rf::f4 t = im.n[y][x];
    return in ? vec4{t.x, t.y, t.z, t.w} : vec4{0.0f, 0.0f, 0.0f, 0.0f};
}
RFG void imageStore(const image2D<BoxPx>& im, ivec2, vec4 v) { im.value = v; }

template <class W, int R> struct WinPx {};
template <class W, int R> struct image2D<WinPx<W, R>> {
    const W* w;
    mutable vec4 value;
};
template <class W, int R> RFG ivec2 imageSize(const image2D<WinPx<W, R>>&) { return ivec2{2 * R + 1, 2 * R + 1}; }
template <class W, int R> RFG vec4 imageLoad(const image2D<WinPx<W, R>>& im, ivec2 q)
{
    int dx = q.x - R, dy = q.y - R;
    dx = dx < -R ? -R : (dx > R ? R : dx);
    dy = dy < -R ? -R : (dy > R ? R : dy);
    const rf::f4 t = im.w->at(dx, dy);
    return vec4{t.x, t.y, t.z, t.w};
}
template <class W, int R> RFG void imageStore(const image2D<WinPx<W, R>>& im, ivec2, vec4 v) { im.value = v; }

RFG void barrier() { __syncthreads(); }
RFG void memoryBarrier() { __threadfence(); }
RFG void memoryBarrierShared() { __threadfence_block(); }
RFG void memoryBarrierImage() { __threadfence(); }
RFG void memoryBarrierBuffer() { __threadfence(); }
RFG void groupMemoryBarrier() { __threadfence_block(); }

// One invocation per thread.  SH = the translated shader (`RfgShader` of its namespace, a template over the texel format), P = rf::PxF32 /
// rf::PxU8, I = its Info:
//   I::LX, LY, LZ   local_size of the file
//   I::GROUPED      the file uses workgroup built-ins, shared variables or barrier(): workgroups are exactly the file's, dispatched
//                   as the reference does; otherwise only gl_GlobalInvocationID is observable and the launch takes 64 x 4 threads
//                   per workgroup over the same set of invocations (a wave then reads 64 adjacent texels of a row = 1 KiB
//                   rgba32f), in XCD-contiguous order
//   I::NIMG, NBUF, UBO   images (in the order of the shader's `bind`), storage buffers, bytes of uniform data
// Args: GlslFrame, then NIMG x GlslImage, NBUF x pointer, UBO bytes (all 8-byte aligned; the host packs the same way).
template <class I> struct GlslArgs {
    GlslFrame f;
    GlslImage img[I::NIMG > 0 ? I::NIMG : 1];
    void* buf[I::NBUF > 0 ? I::NBUF : 1];
    unsigned char ubo[I::UBO > 0 ? (I::UBO + 7) / 8 * 8 : 8];
};

template <template <class> class SH, class P, class I, bool BUF>
__global__ __launch_bounds__(I::GROUPED ? I::LX * I::LY * I::LZ : 256) void glsl_node_kernel(GlslArgs<I> A)
{
    typedef SH<GPx<P, BUF>> S;
    S s;
    uvec3 wg, lid;
    if constexpr (I::GROUPED) {
        const unsigned q = blockIdx.x;
        if (q >= (unsigned)(A.f.groups_x * A.f.groups_y)) return;
        wg = uvec3{q % (unsigned)A.f.groups_x, q / (unsigned)A.f.groups_x, 0u};
        const unsigned t = threadIdx.x;
        lid = uvec3{t % (unsigned)I::LX, (t / (unsigned)I::LX) % (unsigned)I::LY, t / (unsigned)(I::LX * I::LY)};
        // a strip: workgroups none of whose rows this launch writes do not run (a workgroup-uniform decision)
        const int gy0 = (int)wg.y * I::LY;
        if (gy0 + I::LY <= A.f.y0 || gy0 >= A.f.y1) return;
    } else {
        // the same invocations -- x < groups_x * LX, y < groups_y * LY -- in tiles of 64 x 4, every XCD a contiguous range of tiles in
        // raster order (workgroups are dealt round-robin over the 8 XCDs)
        const unsigned tiles_x = ((unsigned)(A.f.groups_x * I::LX) + 63u) / 64u;
        const unsigned y_first = (unsigned)A.f.y0 & ~3u;
        const unsigned tiles_y = ((unsigned)A.f.y1 - y_first + 3u) / 4u;
        const unsigned per_xcd = gridDim.x >> 3;
        const unsigned q = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
        unsigned gx, gy;
        if (A.f.ring > 0) {
            // the border ring of rows y0 .. y1: top rows, bottom rows (whole rows), then the `ring` first and last columns of the rows between
            const int R = A.f.ring, W = A.f.W, H = A.f.H;
            const int t0 = A.f.y0 > 0 ? A.f.y0 : 0, t1 = A.f.y1 < R ? A.f.y1 : R, nt = t1 > t0 ? t1 - t0 : 0;
            const int b0 = A.f.y0 > H - R ? A.f.y0 : H - R, b1 = A.f.y1 < H ? A.f.y1 : H, nb = b1 > b0 ? b1 - b0 : 0;
            const int m0 = A.f.y0 > R ? A.f.y0 : R, m1 = A.f.y1 < H - R ? A.f.y1 : H - R, nm = m1 > m0 ? m1 - m0 : 0;
            long idx = (long)q * 256 + (long)threadIdx.x;
            if (idx < (long)nt * W) { gy = (unsigned)(t0 + (int)(idx / W)); gx = (unsigned)(idx % W); }
            else if ((idx -= (long)nt * W) < (long)nb * W) { gy = (unsigned)(b0 + (int)(idx / W)); gx = (unsigned)(idx % W); }
            else if ((idx -= (long)nb * W) < (long)nm * 2 * R) { gy = (unsigned)(m0 + (int)(idx / (2 * R))); const int c = (int)(idx % (2 * R)); gx = (unsigned)(c < R ? c : W - 2 * R + c); }
            else return;
        } else {
        if (q >= tiles_x * tiles_y) return;
        gx = (q % tiles_x) * 64u + (threadIdx.x & 63u);
        gy = y_first + (q / tiles_x) * 4u + (threadIdx.x >> 6);
        }
        if (gx >= (unsigned)(A.f.groups_x * I::LX) || gy >= (unsigned)(A.f.groups_y * I::LY)) return;
        wg = uvec3{gx / (unsigned)I::LX, gy / (unsigned)I::LY, 0u};
        lid = uvec3{gx % (unsigned)I::LX, gy % (unsigned)I::LY, 0u};
    }
    s.gl_WorkGroupID = wg;
    s.gl_LocalInvocationID = lid;
    s.gl_NumWorkGroups = uvec3{(unsigned)A.f.groups_x, (unsigned)A.f.groups_y, 1u};
    s.gl_GlobalInvocationID = wg * uvec3{(unsigned)I::LX, (unsigned)I::LY, (unsigned)I::LZ} + lid;
    s.gl_LocalInvocationIndex = lid.z * (unsigned)(I::LX * I::LY) + lid.y * (unsigned)I::LX + lid.x;
    s.rfg_bind(A.f, A.img, A.buf, A.ubo);
    if constexpr (!I::GROUPED) {
        if ((int)s.gl_GlobalInvocationID.y < A.f.y0 || (int)s.gl_GlobalInvocationID.y >= A.f.y1) return;
    }
    s.main();
}

}  // namespace rfglsl
#endif
