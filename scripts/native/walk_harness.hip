// Exploration (GPU box): ONE stream kernel instantiation launched by hand with a chosen walk geometry -- compiled on the box in
// seconds, so that variants of rf_stream_dev.h (sed on a copy, -D switches) can be bisected without rebuilding the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I reforge_amd/csrc -I include scripts/native/walk_harness.hip -o /tmp/wh
//   /tmp/wh <W> <H> <rows_per_chunk> <unit> <window> [repeat]
// Prints whether the dynamic launch equals the static one (unit = 0) bit for bit, and the walks taken over.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rf_stream_dev.h"
using namespace rf;

#ifndef HARNESS_R
#define HARNESS_R 15
#endif
#ifndef HARNESS_PF
#define HARNESS_PF 2
#endif
#ifndef HARNESS_PX
#define HARNESS_PX PxF32
#endif
typedef StreamArgs<StHTap<HARNESS_R>, StVTap<HARNESS_R>> Args;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 333, H = argc > 2 ? atoi(argv[2]) : 700, rpc = argc > 3 ? atoi(argv[3]) : 260;
    const int unit = argc > 4 ? atoi(argv[4]) : 8, window = argc > 5 ? atoi(argv[5]) : 16, repeat = argc > 6 ? atoi(argv[6]) : 1;
    constexpr int BPP = HARNESS_PX::BPP, RH = HARNESS_R, VALID = 64 - 2 * RH;
    const size_t pitch = ((size_t)W * BPP + 255) / 256 * 256, bytes = pitch * H;
    char *src, *dst0, *dst1;
    unsigned long long* ws;
    const size_t ws_bytes = 4 << 20;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst0, bytes)); CK(hipMalloc(&dst1, bytes)); CK(hipMalloc(&ws, ws_bytes));
    std::vector<unsigned char> h(bytes);
    unsigned s = 12345;
    for (size_t i = 0; i < bytes; i += 4) { s = s * 1664525u + 1013904223u; const float f = (float)(s >> 8) / 16777216.0f; if (BPP == 16) std::memcpy(&h[i], &f, 4); else std::memcpy(&h[i], &s, 4); }
    CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemset(ws, 0, ws_bytes)); CK(hipMemset(dst0, 0, bytes)); CK(hipMemset(dst1, 0xff, bytes));
    Args A;
    std::memset(&A, 0, sizeof(A));
    A.src = src; A.src_pitch = pitch; A.dst_pitch = pitch;
    A.W = W; A.row_lo = 0; A.row_hi = H - 1; A.y0 = 0; A.y1 = H; A.rows_per_chunk = rpc < H ? rpc : H;
    A.n_strips = (W + VALID - 1) / VALID;
    const int groups = (A.n_strips + kWavesPerBlock - 1) / kWavesPerBlock, chunks = (H + A.rows_per_chunk - 1) / A.rows_per_chunk;
    A.n_work = groups * chunks; A.alternate = getenv("HARNESS_ALT") ? 1 : 0; A.chunks_a = chunks;
    float wsum = 0; float w[HARNESS_R + 1];
    for (int i = 0; i <= HARNESS_R; ++i) { w[i] = 1.0f / (1 + i); wsum += (i ? 2 : 1) * w[i]; }
    for (int i = 0; i <= HARNESS_R; ++i) { A.params.p.w[i] = v2f{w[i] / wsum, w[i] / wsum}; A.params.rest.p.w[i] = v2f{w[i] / wsum, w[i] / wsum}; }
    const unsigned grid = (unsigned)((A.n_work + 7) / 8 * 8);
    std::printf("W %d H %d rpc %d unit %d window %d: %d strips, %d groups, %d chunks, %d workgroups, grid %u\n", W, H, A.rows_per_chunk, unit, window, A.n_strips, groups, chunks, A.n_work, grid);
    // static
    A.dst = dst0;
    hipLaunchKernelGGL((stream_kernel<HARNESS_PX, HARNESS_PF, 1, StHTap<HARNESS_R>, StVTap<HARNESS_R>>), dim3(grid), dim3(64 * kWavesPerBlock), 0, 0, A);
    CK(hipDeviceSynchronize());
    std::printf("static done\n"); std::fflush(stdout);
    A.dst = dst1; A.unit = unit; A.ws = ws; A.steal_window = window; A.stat_word = (int)(ws_bytes / 8) - 1;
    for (int r = 0; r < repeat; ++r) {
        hipLaunchKernelGGL((stream_kernel<HARNESS_PX, HARNESS_PF, 1, StHTap<HARNESS_R>, StVTap<HARNESS_R>>), dim3(grid), dim3(64 * kWavesPerBlock), 0, 0, A);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned char> a(bytes), b(bytes);
    CK(hipMemcpy(a.data(), dst0, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dst1, bytes, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int y = 0; y < H; ++y) if (std::memcmp(&a[y * pitch], &b[y * pitch], (size_t)W * BPP)) { if (!bad) std::printf("first different row %d\n", y); ++bad; }
    std::vector<unsigned long long> words(ws_bytes / 8);
    CK(hipMemcpy(words.data(), ws, ws_bytes, hipMemcpyDeviceToHost));
    size_t nonempty = 0;
    for (size_t i = 0; i + 1 < words.size(); ++i) { const unsigned lo = (unsigned)words[i]; if ((lo & 0xffff) < (lo >> 16)) ++nonempty; }
    std::printf("dynamic: %zu different rows, walks taken %llu, words left non-empty %zu\n", bad, words.back(), nonempty);
    return bad || nonempty ? 1 : 0;
}
