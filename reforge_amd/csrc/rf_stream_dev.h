// rf_stream_dev.h -- DEVICE code of the streaming stage-pipeline kernel (see rf_stream.hip for the design notes).
//
// This header is compiled twice: ahead of time by hipcc as part of rf_stream.hip (the kernel catalogue), and at
// graph creation by hiprtc (rf_jit.cpp), which instantiates stream_kernel<> for a stage list the catalogue lacks --
// the counterpart of the reference compiling a node's shader when the graph is built (src/vulkan/shader.rs:29-93).
// It therefore includes nothing but rf_device.h and uses no host-side type.
#pragma once
#include "rf_device.h"

namespace rf {

// ---------------------------------------------------------------------------------
// Per-lane context of a streaming wave.  A wave's strip is 64*T texels wide: lane l owns the
// T texels at strip positions l, l + 64, ... (texel j of the lane = position l + 64 j), so every
// global access and every LDS row access of a wave instruction is 64 consecutive texels -- one
// fully coalesced 1 KiB segment for rgba32f -- whatever T is.  T = 2 halves the share of halo
// lanes (2 RH of 128 instead of 2 RH of 64) and the per-row scalar work per texel.
// ---------------------------------------------------------------------------------
template <int T> struct Tex { f4 v[T]; };

template <int T> RF_DEV Tex<T> tex_zero()
{
    Tex<T> z;
#pragma unroll
    for (int j = 0; j < T; ++j) z.v[j] = f4_zero();
    return z;
}

template <int T> struct Lane {
    int lane;   // 0..63
    int x0;     // frame column of strip position 0 (may be negative: left halo)
    int W;
    f4* lds;    // wave-private LDS rows, 64*T texels each
    RF_DEV int pos(int j) const { return lane + 64 * j; }          // strip position of the lane's texel j
    RF_DEV int col(int j) const { return x0 + lane + 64 * j; }     // its frame column (may lie outside [0,W) in the halo)
    // LDS slot holding column clamp(col(j)+dx) -- clamp-to-edge at the frame border, and
    // kept inside the wave's row for the halo lanes (whose results are discarded)
    RF_DEV int nbr(int j, int dx) const
    {
        int c = min(max(col(j) + dx, 0), W - 1) - x0;
        return min(max(c, 0), 64 * T - 1);
    }
};

// LDS operations of one wave execute in issue order, so a wave-private exchange
// needs no s_barrier: only the compiler has to be told not to reorder.
RF_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------
// Source: the wave's input rows, streamed global -> LDS by LDS-DMA (global_load_lds_*:
// no VGPR destination) into a wave-private ring of PF row slots, PF rows ahead.
//
// Why hand-written: with compiler-visible loads hipcc drains EVERY load in flight
// (s_waitcnt vmcnt(0)) at each use, because loads and stores share vmcnt on gfx9 and its
// wait-count pass treats mixed pending events as out of order.  The DMA is issued from an
// asm statement (invisible to that pass) and waited for with a COUNTED vmcnt: vector
// memory operations retire in issue order, and a wave issues exactly T DMAs per input
// row and T stores per output row, in a fixed program order (see wait_row).
// ---------------------------------------------------------------------------------
template <int N> RF_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int T> struct Sink {
    char* dst;          // address of the next output row (wave-uniform: advances on the scalar unit)
    ptrdiff_t pitch;    // negative when the wave walks bottom-up
    unsigned xoff[T];   // byte offset of the lane's texel j in a row
    bool lane_ok[T];    // the lane owns an output texel at j
    int row;            // next output row
    int first_store;    // iteration of the first store, -1 before it (wave-uniform)
};

template <class Px, int PF, int T> struct Source {
    static_assert(PF >= 2, "the ring needs at least two slots");
    static constexpr int SLOTS = PF;
    static constexpr int SLOT_BYTES = 64 * T * Px::BPP;
    const char* src;      // address of local row 0 (wave-uniform)
    unsigned xoff[T];     // byte offset of the column of the lane's texel j (clamped to the frame)
    mutable const char* cur;   // address of the next row to issue: rows are issued in order, the pointer advances on the scalar unit
    ptrdiff_t pitch;
    int a0, n0;           // first source row, number of source rows
    unsigned lds_base;    // LDS byte address of slot 0 (wave-uniform)
    const char* ring;     // the same ring through a generic pointer

    RF_DEV const char* slot(int r) const { return ring + (size_t)(r % SLOTS) * SLOT_BYTES; }

    // DMA source row r into slot r % PF.  Program order inside iteration `it` is
    //   [first stage consumes row it] -> issue(it+PF) -> wait_row(it+1) -> ... -> store
    // The slot being refilled is the one row `it` was read from (ds_read, one iteration ago).
    // The DMA's data arrives through the memory path, not through the LDS instruction queue, and
    // the compiler is free to sink the first USE of that ds_read -- and with it the only
    // s_waitcnt lgkmcnt that proves the read has executed -- below this asm.  An L2 hit (the
    // neighbour chunk has just fetched the same halo rows) then overtakes a ds_read still queued
    // behind other waves' LDS traffic and the stage computes on the NEXT row's texels: wrong
    // first rows of a walk, seen only on busy chips (tests/test_gpu_fullsize.py::
    // test_random_graphs_1080p_whole_frame).  Hence the lgkmcnt(0) in front of the DMA: by then
    // the taps of the row have normally been consumed and the wait is free.
    RF_DEV void issue(int r) const
    {
        const unsigned dst0 = lds_base + (unsigned)(r % SLOTS) * (unsigned)SLOT_BYTES;
        const char* g = cur;
        cur = g + pitch;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const unsigned dst = dst0 + (unsigned)(j * 64 * Px::BPP);
            unsigned keep;
            // RF_SBASE (rf_device.h): the row address may have been reloaded by a vector instruction right in front of the asm
            if constexpr (Px::BPP == 16)
                asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_SBASE "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[off], vcc" RF_LOAD_MOD "\n\ts_mov_b32 m0, %[keep]"
                             : [keep] "=&s"(keep) : [off] "v"(xoff[j]), [lds] "s"(dst), [base] "s"(g) : "memory", "vcc");
            else
                asm volatile("s_waitcnt lgkmcnt(0)\n\t" RF_SBASE "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dword %[off], vcc" RF_LOAD_MOD "\n\ts_mov_b32 m0, %[keep]"
                             : [keep] "=&s"(keep) : [off] "v"(xoff[j]), [lds] "s"(dst), [base] "s"(g) : "memory", "vcc");
        }
    }
    RF_DEV void prologue() const
    {
        cur = src + (ptrdiff_t)a0 * pitch;
        for (int r = 0; r < PF && r < n0; ++r) issue(r);
    }
    // Wait until row r has landed, leaving younger operations in flight.  Younger than
    // row r's DMAs at this point: the DMAs of rows r+1 .. r+PF-1 (when they exist) and, once
    // the pipeline emits a row per iteration, the stores of the PF-1 iterations in between
    // (T of each per row).
    RF_DEV void wait_row(int r, const Sink<T>& k) const
    {
        if (n0 - 1 - r >= PF - 1) {
            if (k.first_store >= 0 && k.first_store <= r - PF) wait_vmcnt<(2 * PF - 2) * T>();
            else wait_vmcnt<(PF - 1) * T>();
        } else {
            wait_vmcnt<0>();
        }
    }
};

// what the first stage is handed each iteration, fetched from the ring one iteration
// ahead so the LDS latency hides behind the previous row's arithmetic
template <class Px, int T> struct OwnFeed {      // the lane's own texels
    typename Px::Raw nxt[T];
    template <class Src> RF_DEV void fetch(const Src& s, int r, const Lane<T>& L) { fetch_at(s.slot(r), L); }
    RF_DEV void fetch_at(const char* slot, const Lane<T>& L)
    {
#pragma unroll
        for (int j = 0; j < T; ++j) nxt[j] = *reinterpret_cast<const typename Px::Raw*>(slot + (size_t)L.pos(j) * Px::BPP);
    }
    RF_DEV Tex<T> own() const
    {
        Tex<T> o;
#pragma unroll
        for (int j = 0; j < T; ++j) o.v[j] = Px::decode(nxt[j]);
        return o;
    }
};
template <int R, int T> struct TapFeed {         // rgba32f: the 2R+1 horizontal taps, straight from the DMA ring
    f4 t[T][2 * R + 1];
    // The ring row already holds clamp-to-edge columns (the DMA reads column clamp(x)), so tap i of position p is
    // simply slot p + i: ONE address per texel (position p - R; the taps are instruction offsets) instead of 2R+1.
    // Positions within R of the strip's ends -- halo lanes, results discarded -- are pulled inside the row.
    template <class Src> RF_DEV void fetch(const Src& s, int r, const Lane<T>& L) { fetch_at(s.slot(r), L); }
    RF_DEV void fetch_at(const char* slot, const Lane<T>& L)
    {
        const f4* row = reinterpret_cast<const f4*>(slot);
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const f4* base = row + min(max(L.pos(j) - R, 0), 64 * T - 1 - 2 * R);
#pragma unroll
#ifdef RF_EXPERIMENT_NO_EXCHANGE     // timing-only build (scripts/mk_variant.sh): every tap reads the centre texel -- WRONG results
            for (int i = 0; i <= 2 * R; ++i) t[j][i] = base[R];
#else
            for (int i = 0; i <= 2 * R; ++i) t[j][i] = base[i];
#endif
        }
    }
    RF_DEV Tex<T> own() const
    {
        Tex<T> o;
#pragma unroll
        for (int j = 0; j < T; ++j) o.v[j] = t[j][R];
        return o;
    }
};

// ---------------------------------------------------------------------------------
// Row stages.  advance() is called once per row entering the stage:
//   v      the row's texels for this lane (undefined when !real)
//   real   a new input row; false = the newest row repeated (clamp-to-edge below the frame)
//   first  the stage's first row: it primes the whole window (clamp-to-edge above the frame,
//          or rows that are shifted out again before anything is emitted)
//   emit   the window's centre row is wanted downstream (wave-uniform, from the schedule)
// ---------------------------------------------------------------------------------
struct NoState {};

// horizontal taps of the separable gaussian: sum_i w[|i|] * in[x+i], ascending i
template <int R> struct StHTap {
    static constexpr int RV = 0, RH = R, LDS_ROWS = (R > 0) ? 1 : 0;
    struct Params { v2f w[R + 1]; };   // each weight twice: the operand pair of a packed fma (see fma4)
    template <class Px, int T> using State = NoState;
    // as the FIRST stage of an rgba32f pipeline the taps come straight from the DMA ring
    // (up to radius 7: beyond that 2R+1 prefetched taps crowd the vertical window out of the register
    // file -- radius 10 at 4K: 209 us with the prefetch, 119 us through the LDS exchange)
    template <class Px, int T> using Feed = typename std::conditional<(Px::QUANT || R > 7 || T > 1), OwnFeed<Px, T>, TapFeed<R, T>>::type;
    template <int T> RF_DEV static Tex<T> from_taps(const Params& p, const TapFeed<R, T>& f)
    {
        Tex<T> o;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f4 acc = f4_zero();
#pragma unroll
            for (int i = -R; i <= R; ++i) acc = fma4(p.w[i < 0 ? -i : i], f.t[j][i + R], acc);
            o.v[j] = acc;
        }
        return o;
    }
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, NoState&, const Lane<T>& L, f4* lds, const Tex<T>& v, bool, bool, bool, Tex<T>& out)
    {
#ifndef RF_EXPERIMENT_NO_EXCHANGE
        if constexpr (R > 0) {
#pragma unroll
            for (int j = 0; j < T; ++j) lds[L.pos(j)] = v.v[j];
            wave_sync();
        }
#endif
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f4 acc = f4_zero();
#pragma unroll
            for (int i = -R; i <= R; ++i) {
#ifdef RF_EXPERIMENT_NO_EXCHANGE
                f4 t = v.v[j];
#else
                f4 t = (i == 0) ? v.v[j] : lds[L.nbr(j, i)];
#endif
                acc = fma4(p.w[i < 0 ? -i : i], t, acc);
            }
            out.v[j] = acc;
        }
    }
};

// vertical taps: sum_j w[|j|] * tmp[y+j], ascending j.
//
// Walking top-down the rows arrive in ascending order, which IS the tap order of every output row:
// the stage then keeps one running sum per pending output row instead of a window of input rows
// (SCATTER form).  Row r contributes  acc[y] = fma(w|r-y|, tmp[r], acc[y])  to the 2R+1 outputs it
// reaches; the oldest of them, y = r - R, takes its last tap and is emitted.  Each output still
// sums its taps in ascending j from 0 -- bit-identical to the window form -- but the slide of the
// pending rows is done by the fma itself (destination = the slot one up from its addend), where a
// window of rows has to be shifted with moves: 2 x 2R v_mov_b64 per row and texel.
// Walking bottom-up (REV) the rows arrive in DESCENDING order, so the sums cannot be formed on
// arrival; those walks keep the window of rows (GATHER form).
template <int R> struct StVTap {
    static constexpr int RV = R, RH = 0, LDS_ROWS = 0;
    struct Params { v2f w[R + 1]; };
    // scatter: acc[k] = running sum of output row (newest - R + 1 + k), k = 0 .. 2R-1; last = newest real row (bottom-edge flush)
    // gather:  win[i] = input row (newest - 2R + i)
    template <class Px, int T> struct State { Tex<T> win[2 * R + 1]; };
    template <class Px, int T> using Feed = OwnFeed<Px, T>;

    // running sums 0 .. 2R-2 slide down by one while taking their tap (compile-time recursion: every weight
    // index is a constant for the front end already, so the parameter block stays in scalar registers)
    template <int K, int T> RF_DEV static void slide(const Params& p, Tex<T>* acc, const f4& v, int t)
    {
        if constexpr (K + 1 < 2 * R) {
            constexpr int j = R - 1 - K;                                       // row newest-R+1+K takes tap j
            acc[K].v[t] = fma4(p.w[j < 0 ? -j : j], v, acc[K + 1].v[t]);
            slide<K + 1, T>(p, acc, v, t);
        }
    }
    template <int T> RF_DEV static void scatter_row(const Params& p, Tex<T>* acc, const Tex<T>& v, Tex<T>& out)
    {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            out.v[t] = fma4(p.w[R], v.v[t], acc[0].v[t]);                       // j = +R: the last tap of row newest-R
            slide<0, T>(p, acc, v.v[t], t);
            acc[2 * R - 1].v[t] = fma4(p.w[R], v.v[t], f4_zero());              // j = -R: the first tap of row newest+R
        }
    }
    // KEEP: the row may be the stage's last real one (tail and generic loop phases): remember it for the flush
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, State<Px, T>& s, const Lane<T>&, f4*, const Tex<T>& v, bool real, bool first, bool emit, Tex<T>& out)
    {
        if constexpr (R == 0) {
            out = tex_zero<T>();
#pragma unroll
            for (int t = 0; t < T; ++t) out.v[t] = fma4(p.w[0], v.v[t], f4_zero());
        } else if constexpr (!REV) {
            // s.win[0 .. 2R-1] are the running sums, s.win[2R] the newest real row
            if (first) {
                // clamp-to-edge above the stage's first row: as if 2R more copies of it had arrived before
#pragma unroll
                for (int k = 0; k < 2 * R; ++k) s.win[k] = tex_zero<T>();
                Tex<T> dummy;
#pragma unroll
                for (int n = 0; n < 2 * R; ++n) scatter_row<T>(p, s.win, v, dummy);
            }
            Tex<T> in;                        // by value: a reference picked between two objects would pin both to memory
#pragma unroll
            for (int t = 0; t < T; ++t) in.v[t] = real ? v.v[t] : s.win[2 * R].v[t];
            Tex<T> o;
            scatter_row<T>(p, s.win, in, o);
            if (emit) out = o;
            if constexpr (KEEP) {
                if (real) s.win[2 * R] = v;
            }
        } else {
            if (first) {                     // the first row primes the whole window (it is always a real row)
#pragma unroll
                for (int i = 0; i <= 2 * R; ++i) s.win[i] = v;
            } else {
#pragma unroll
                for (int i = 0; i < 2 * R; ++i) s.win[i] = s.win[i + 1];
                if (real) s.win[2 * R] = v;
            }
            if (emit) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    f4 acc = f4_zero();
#pragma unroll
                    // taps are accumulated in ascending FRAME row order; walking bottom-up the window
                    // holds the rows the other way round
                    for (int j = -R; j <= R; ++j) acc = fma4(p.w[j < 0 ? -j : j], s.win[R - j].v[t], acc);
                    out.v[t] = acc;
                }
            }
        }
    }
};

// colour grade point op
struct StGrade {
    static constexpr int RV = 0, RH = 0, LDS_ROWS = 0;
    struct Params { float slope, offset, saturation; };
    template <class Px, int T> using State = NoState;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    RF_DEV static float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
    RF_DEV static f4 grade(const Params& p, f4 c)
    {
        float tr = fmaf(c.x, p.slope, p.offset);
        float tg = fmaf(c.y, p.slope, p.offset);
        float tb = fmaf(c.z, p.slope, p.offset);
        float luma = fmaf(0.0722f, tb, fmaf(0.7152f, tg, 0.2126f * tr));
        return make_float4(clamp01(fmaf(p.saturation, tr - luma, luma)), clamp01(fmaf(p.saturation, tg - luma, luma)),
                           clamp01(fmaf(p.saturation, tb - luma, luma)), c.w);
    }
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, NoState&, const Lane<T>&, f4*, const Tex<T>& c, bool, bool, bool, Tex<T>& out)
    {
#pragma unroll
        for (int j = 0; j < T; ++j) out.v[j] = grade(p, c.v[j]);
    }
};

// 3x3 sharpen cross [0,s,0; s,c,s; 0,s,0], taps in ascending (y outer, x inner) order.
// The horizontal neighbours of a row are fetched through LDS when the row ARRIVES and
// are first used one iteration later, when that row is the centre: the LDS round trip
// hides behind a whole iteration instead of stalling the wave.
struct StCross3 {
    static constexpr int RV = 1, RH = 1, LDS_ROWS = 1;
    // each weight twice, like the gaussian taps: the operand pair of a packed fma, ONE aligned 8-byte scalar load each.  As two
    // adjacent floats, the older compiler PyTorch bundles (the hiprtc a process that imported torch gets) widened the loads into
    // overlapping vector loads and pinned the block to scratch -- scratch loads inside the counted-vmcnt loops
    // (tests/test_jit_isa.py found it; stream_prepare refuses such a kernel, so the cost was a split chain, not wrong texels)
    struct Params { v2f wc, ws; };
    // bottom-up walks keep a window: n, c = rows y-1, y (in walk order), cw, ce = y's left/right neighbours.
    // Top-down walks keep RUNNING SUMS instead (as StVTap does): n = the sum of output row y so far (its N, W, C, E
    // taps), c = the N tap of output row y+1, cw = the newest real row (bottom-edge flush; KEEP phases only) -- the same
    // five fmas in the same order, and no register moves per row.
    template <class Px, int T> struct State { Tex<T> n, c, cw, ce; };
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <int T> RF_DEV static void exchange(const Lane<T>& L, f4* lds, const Tex<T>& v, Tex<T>& w, Tex<T>& e)
    {
#ifdef RF_EXPERIMENT_NO_EXCHANGE
        w = v;
        e = v;
        return;
#endif
#pragma unroll
        for (int j = 0; j < T; ++j) lds[L.pos(j)] = v.v[j];
        wave_sync();
#pragma unroll
        for (int j = 0; j < T; ++j) {
            w.v[j] = lds[L.nbr(j, -1)];
            e.v[j] = lds[L.nbr(j, +1)];
        }
    }
    // row v becomes the centre row: its W, C, E taps go on top of the N tap already summed in `north`
    template <int T> RF_DEV static void centre_row(const Params& p, const Lane<T>& L, f4* lds, const Tex<T>& v, const Tex<T>& north, Tex<T>& sum, Tex<T>& next_north)
    {
        Tex<T> w, e;
        exchange(L, lds, v, w, e);
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f4 acc = fma4(p.ws, w.v[j], north.v[j]);
            acc = fma4(p.wc, v.v[j], acc);
            sum.v[j] = fma4(p.ws, e.v[j], acc);
            next_north.v[j] = fma4(p.ws, v.v[j], f4_zero());
        }
    }
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, State<Px, T>& s, const Lane<T>& L, f4* lds, const Tex<T>& v, bool real, bool first, bool emit, Tex<T>& out)
    {
        if constexpr (!REV) {
            if (first) {                 // clamp-to-edge above the stage's first row: it is its own northern neighbour
                Tex<T> north;
#pragma unroll
                for (int j = 0; j < T; ++j) north.v[j] = fma4(p.ws, v.v[j], f4_zero());
                centre_row<T>(p, L, lds, v, north, s.n, s.c);
                if constexpr (KEEP) s.cw = v;
                return;
            }
            if (emit) {
#pragma unroll
                for (int j = 0; j < T; ++j) out.v[j] = fma4(p.ws, real ? v.v[j] : s.cw.v[j], s.n.v[j]);     // the S tap closes the sum
            }
            if (real) {
                const Tex<T> north = s.c;
                centre_row<T>(p, L, lds, v, north, s.n, s.c);
                if constexpr (KEEP) s.cw = v;
            }
            return;
        }
        if (first) {                     // window = [v, v, (next row)]
            s.n = v;
            s.c = v;
            exchange(L, lds, v, s.cw, s.ce);
            return;
        }
        const Tex<T> below = real ? v : s.c;
        if (emit) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f4 acc = f4_zero();
                // frame order N, W, C, E, S: walking bottom-up the older row is the one BELOW
                acc = fma4(p.ws, below.v[j], acc);
                acc = fma4(p.ws, s.cw.v[j], acc);
                acc = fma4(p.wc, s.c.v[j], acc);
                acc = fma4(p.ws, s.ce.v[j], acc);
                acc = fma4(p.ws, s.n.v[j], acc);
                out.v[j] = acc;
            }
        }
        s.n = s.c;
        if (real) {
            s.c = v;
            exchange(L, lds, v, s.cw, s.ce);
        }
    }
};

// node boundary inside a fused chain: the store + load the unfused graph performs
// (UNORM8 re-quantisation for rgba8, nothing for rgba32f)
struct StNodeEnd {
    static constexpr int RV = 0, RH = 0, LDS_ROWS = 0;
    struct Params {};
    template <class Px, int T> using State = NoState;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params&, NoState&, const Lane<T>&, f4*, const Tex<T>& v, bool, bool, bool, Tex<T>& out)
    {
#pragma unroll
        for (int j = 0; j < T; ++j) out.v[j] = Px::requant(v.v[j]);
    }
};

// A USER stage: a filter type that is a FILE, `{shader_path}/{type}.stage.hip` -- the counterpart of the reference's
// `{shader_path}/{type}.comp` (src/config/config.rs:59-75), compiled when a graph that names it is created
// (src/vulkan/shader.rs:29-93) and recompiled when it changes (src/render.rs:225-249).  The file declares
//     struct Params { float a; int b; bool c; };     // the config's parameter names (uniform members, render.rs:169-185)
//     static constexpr int RADIUS = 0 | 1;           // point op | 3 x 3 neighbourhood
//     RF_STAGE f4 apply(const Params& p, f4 c);                       // RADIUS 0
//     RF_STAGE f4 apply(const Params& p, const f4 (&n)[3][3]);        // RADIUS 1: n[dy + 1][dx + 1], clamp-to-edge
// rf_user.cpp wraps the text into namespace rfuser::u_<hash> with a `Stage` type (U below) and rf_jit.cpp appends it to this
// source; a user stage is a row stage like any other, so user nodes FUSE with the built-in ones.
// The 3 x 3 form keeps a window of the two previous rows with their west / east neighbours (fetched through the wave's LDS
// row when a row arrives, like StCross3); rows are handed to apply() in FRAME order whatever the walk direction.
template <class U> struct StUser {
    static constexpr int RV = U::R, RH = U::R, LDS_ROWS = U::R > 0 ? 1 : 0;
    struct Params { typename U::P p; };
    struct Row3 { f4 w, c, e; };
    template <class Px, int T> struct State { Row3 prev[T], cur[T]; };      // walk order: prev = row y-1, cur = row y
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <int T> RF_DEV static void exchange(const Lane<T>& L, f4* lds, const Tex<T>& v, Row3* r)
    {
#pragma unroll
        for (int j = 0; j < T; ++j) lds[L.pos(j)] = v.v[j];
        wave_sync();
#pragma unroll
        for (int j = 0; j < T; ++j) {
            r[j].w = lds[L.nbr(j, -1)];
            r[j].c = v.v[j];
            r[j].e = lds[L.nbr(j, +1)];
        }
    }
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, State<Px, T>& s, const Lane<T>& L, f4* lds, const Tex<T>& v, bool real, bool first, bool emit, Tex<T>& out)
    {
        if constexpr (U::R == 0) {
#pragma unroll
            for (int j = 0; j < T; ++j) out.v[j] = U::point(p.p, v.v[j]);
        } else {
            Row3 nw[T];
            if (real) exchange<T>(L, lds, v, nw);
            if (first) {                     // clamp-to-edge before the stage's first row: it is its own predecessor
#pragma unroll
                for (int j = 0; j < T; ++j) { s.prev[j] = nw[j]; s.cur[j] = nw[j]; }
                return;
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const Row3 next = real ? nw[j] : s.cur[j];          // past the last row: the newest row again
                if (emit) {
                    const Row3& north = REV ? next : s.prev[j];
                    const Row3& south = REV ? s.prev[j] : next;
                    const f4 n[3][3] = {{north.w, north.c, north.e}, {s.cur[j].w, s.cur[j].c, s.cur[j].e}, {south.w, south.c, south.e}};
                    out.v[j] = U::box(p.p, n);
                }
                s.prev[j] = s.cur[j];
                if (real) s.cur[j] = nw[j];
            }
        }
    }
};

// ---------------------------------------------------------------------------------
// Fork / join in ONE launch (the diamond of src/vulkan/pipeline_graph.rs:462-468: two branches read one image
// and a `combination` node joins them).  Between the fork and the join the value that travels down the stage
// chain is a PAIR of rows -- slot 0 = texels [0, T) of a Tex<2T>, slot 1 = [T, 2T) -- and the two branches are
// laid end to end in the stage list, but they run SIDE BY SIDE: a stage of branch K works on slot K and leaves the
// other slot alone, every stage keeps the schedule of ITS slot's rows (Chain: has-bits, needs and last-emission
// iterations are per slot), the fork hands each slot the rows that slot needs, and the branch of the smaller
// vertical radius ends in a delay line (StDelay) so that both slots reach the join with the same frame row.  The
// launch therefore reads pre + max(a, b) + post halo rows -- what a row strip exchanges or over-fetches -- and loses
// pre + max(a, b) + post halo lanes per side.  (Until round 4 the branches ran one BEHIND the other, the idle slot
// riding a delay line per stage: a + b.)  The source rows cross the fabric once for both branches and neither branch
// result is ever stored: the 4K diamond moves 2 image passes instead of 7.
// ---------------------------------------------------------------------------------
template <class S> struct SlotsOf { static constexpr int value = 1; };

template <int T> RF_DEV Tex<T> take_slot(const Tex<2 * T>& v, int k)
{
    Tex<T> o;
#pragma unroll
    for (int j = 0; j < T; ++j) o.v[j] = v.v[k * T + j];
    return o;
}
template <int T> RF_DEV void put_slot(Tex<2 * T>& v, int k, const Tex<T>& a)
{
#pragma unroll
    for (int j = 0; j < T; ++j) v.v[k * T + j] = a.v[j];
}

// a plain stage inside a pair pipeline, before the fork or after the join: slot 0 only
template <class S> struct StSolo {
    static constexpr int RV = S::RV, RH = S::RH, LDS_ROWS = S::LDS_ROWS;
    typedef typename S::Params Params;
    template <class Px, int T> using State = typename S::template State<Px, T>;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, State<Px, T>& s, const Lane<T>& L, f4* lds, const Tex<2 * T>& v, bool real, bool first, bool emit, Tex<2 * T>& out)
    {
        Tex<T> o = tex_zero<T>();
        S::template advance<Px, REV, T, KEEP>(p, s, L, lds, take_slot<T>(v, 0), real, first, emit, o);
        put_slot<T>(out, 0, o);
    }
};
template <class S> struct SlotsOf<StSolo<S>> { static constexpr int value = 2; };

// the fork: both branches start from the same row (Chain gives each slot the rows IT needs: see Chain::fork_bits)
struct StDup {
    static constexpr int RV = 0, RH = 0, LDS_ROWS = 0;
    struct Params {};
    template <class Px, int T> using State = NoState;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params&, NoState&, const Lane<T>&, f4*, const Tex<2 * T>& v, bool, bool, bool, Tex<2 * T>& out)
    {
        put_slot<T>(out, 0, take_slot<T>(v, 0));
        put_slot<T>(out, 1, take_slot<T>(v, 0));
    }
};
template <> struct SlotsOf<StDup> { static constexpr int value = 2; };

// stage S on slot K; the other slot passes (Chain copies it through whether or not this stage has a row this iteration)
template <int K, class S> struct StOn {
    static constexpr int RV = S::RV, RH = S::RH, LDS_ROWS = S::LDS_ROWS;
    typedef typename S::Params Params;
    template <class Px, int T> using State = typename S::template State<Px, T>;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, State<Px, T>& s, const Lane<T>& L, f4* lds, const Tex<2 * T>& v, bool real, bool first, bool emit, Tex<2 * T>& out)
    {
        Tex<T> o = tex_zero<T>();
        S::template advance<Px, REV, T, KEEP>(p, s, L, lds, take_slot<T>(v, K), real, first, emit, o);
        put_slot<T>(out, K, o);
    }
};
template <int K, class S> struct SlotsOf<StOn<K, S>> { static constexpr int value = 2; };

// D rows of delay on slot K: the end of the branch with the smaller vertical radius.  It needs no row above the one it emits
// and D rows below (the rows that push it out): RT = 0, RB = D in the schedule (every other stage: RT = RB = RV).
template <int K, int D> struct StDelay {
    static_assert(D >= 1, "a delay of nothing is no stage");
    static constexpr int RV = 0, RH = 0, LDS_ROWS = 0;
    struct Params {};
    template <class Px, int T> struct State { Tex<T> line[D]; };
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params&, State<Px, T>& s, const Lane<T>&, f4*, const Tex<2 * T>& v, bool, bool, bool, Tex<2 * T>& out)
    {
        put_slot<T>(out, K, s.line[0]);      // the row that arrived D ticks ago (before that: never emitted, see the schedule)
#pragma unroll
        for (int i = 0; i + 1 < D; ++i) s.line[i] = s.line[i + 1];
        s.line[D - 1] = take_slot<T>(v, K);
    }
};
template <int K, int D> struct SlotsOf<StDelay<K, D>> { static constexpr int value = 2; };

// the join: combination, out = fma(mix, b - a, a) per channel (oracle/rf_oracle.c rfo_mix), a = slot 0, b = slot 1
struct StMix {
    static constexpr int RV = 0, RH = 0, LDS_ROWS = 0;
    struct Params { float mix; };
    template <class Px, int T> using State = NoState;
    template <class Px, int T> using Feed = OwnFeed<Px, T>;
    template <class Px, bool REV, int T, bool KEEP> RF_DEV static void advance(const Params& p, NoState&, const Lane<T>&, f4*, const Tex<2 * T>& v, bool, bool, bool, Tex<2 * T>& out)
    {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const f4 a = v.v[j], b = v.v[T + j];
            out.v[j] = make_float4(fmaf(p.mix, b.x - a.x, a.x), fmaf(p.mix, b.y - a.y, a.y), fmaf(p.mix, b.z - a.z, a.z), fmaf(p.mix, b.w - a.w, a.w));
        }
    }
};
template <> struct SlotsOf<StMix> { static constexpr int value = 2; };

template <class... S> struct MaxSlots { static constexpr int value = 1; };
template <class S, class... Rest> struct MaxSlots<S, Rest...> { static constexpr int value = SlotsOf<S>::value > MaxSlots<Rest...>::value ? SlotsOf<S>::value : MaxSlots<Rest...>::value; };

// ---------------------------------------------------------------------------------
// Parameter pack (kernel argument) and the stage chain (per-wave state)
// ---------------------------------------------------------------------------------
// Layout by construction: every stage owns a slot of max(8, sizeof(Params) rounded up to 8) bytes, in stage order,
// closed by one empty 8-byte slot.  The host fills the block as BYTES (rf_stream.hip, param_bytes) -- the same code
// for a catalogue kernel and for one compiled at graph creation, whose stage list no host template ever saw.
template <class... S> struct ParamPack;
template <> struct alignas(8) ParamPack<> {};
template <class S, class... Rest> struct alignas(8) ParamPack<S, Rest...> {
    alignas(8) typename S::Params p;
    alignas(8) ParamPack<Rest...> rest;
};

// which slot of a pair pipeline a stage works on: 0 / 1 for the stages of a branch, -1 for everything else (a plain chain, the
// stages before the fork and after the join -- slot 0 --, the fork and the join themselves)
template <class S> struct BranchOf { static constexpr int value = -1; };
template <int K, class S> struct BranchOf<StOn<K, S>> { static constexpr int value = K; };
template <int K, int D> struct BranchOf<StDelay<K, D>> { static constexpr int value = K; };
// rows a stage needs above / below the row it emits
template <class S> struct RowsAbove { static constexpr int value = S::RV; };
template <class S> struct RowsBelow { static constexpr int value = S::RV; };
template <int K, int D> struct RowsBelow<StDelay<K, D>> { static constexpr int value = D; };
// reach of a stage list: the stages outside the branches add up, of the two branches the larger counts
template <int B, class... S> struct BranchRH { static constexpr int value = 0; };
template <int B, class S, class... Rest> struct BranchRH<B, S, Rest...> { static constexpr int value = (BranchOf<S>::value == B ? S::RH : 0) + BranchRH<B, Rest...>::value; };
template <int B, class... S> struct BranchRV { static constexpr int value = 0; };
template <int B, class S, class... Rest> struct BranchRV<B, S, Rest...> { static constexpr int value = (BranchOf<S>::value == B ? S::RV : 0) + BranchRV<B, Rest...>::value; };
template <class... S> struct SumRH {
    static constexpr int value = BranchRH<-1, S...>::value + (BranchRH<0, S...>::value > BranchRH<1, S...>::value ? BranchRH<0, S...>::value : BranchRH<1, S...>::value);
};
template <class... S> struct SumLDS { static constexpr int value = 0; };
template <class S, class... Rest> struct SumLDS<S, Rest...> { static constexpr int value = S::LDS_ROWS + SumLDS<Rest...>::value; };
template <class... S> struct SumRV {
    static constexpr int value = BranchRV<-1, S...>::value + (BranchRV<0, S...>::value > BranchRV<1, S...>::value ? BranchRV<0, S...>::value : BranchRV<1, S...>::value);
};
template <class... S> struct MaxRV { static constexpr int value = 0; };
template <class S, class... Rest> struct MaxRV<S, Rest...> { static constexpr int value = S::RV > MaxRV<Rest...>::value ? S::RV : MaxRV<Rest...>::value; };
template <class S, class...> struct FirstOf { typedef S type; };

// REV: the wave walks its chunk bottom-up (rows are addressed with a negated pitch, so the
// schedule below is unchanged); stages whose tap order depends on the row direction read it.
//
// The schedule is PER SLOT (NS = 2 in a fork/join pipeline, where the two branches run side by side on the two halves of the
// value): which rows a slot's stream carries (Rows), at which iteration its upstream emitted last (Ticks), and whether it
// carries a row this iteration (bit k of `hb`).  A plain chain has one slot and everything below folds to scalars.
template <int NS> struct Rows { int a[NS], b[NS]; };      // first / last row of each slot's stream
template <int NS> struct Ticks { int t[NS]; };
template <class Px, bool REV, int T, int NS, int LdsIdx, class... S> struct Chain;     // NS: slots of the value between stages (2 in a fork/join pipeline)

// end of the chain: the store
template <class Px, bool REV, int T, int NS, int LdsIdx> struct Chain<Px, REV, T, NS, LdsIdx> {
    RF_DEV void plan_backward(int oa, int ob, int, int, Rows<NS>& in)
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) { in.a[k] = oa; in.b[k] = ob; }
    }
    RF_DEV int plan_forward(const Ticks<NS>& tp) { return tp.t[0]; }
    template <int MODE> RF_DEV void step(unsigned hb, const Tex<T * NS>& v, int it, const Lane<T>&, Sink<T>& k, const ParamPack<>&)
    {
        constexpr bool STEADY = MODE != 0;
        if (STEADY || (hb & 1u)) {
            // the row's values are computed HERE, under the full exec mask: left to itself hipcc sinks the
            // last stage's arithmetic into the exec-masked store block and schedules it there as one
            // serial chain per half texel with an s_nop between dependent packed fmas
#pragma unroll
            for (int j = 0; j < T; ++j) asm volatile("" ::"v"(v.v[j].x), "v"(v.v[j].y), "v"(v.v[j].z), "v"(v.v[j].w));
            // exactly T vector-memory instructions per emitted row: Source::wait_row counts on it
            // (every one of them has at least one active lane: see the strip placement in stream_kernel)
#pragma unroll
            for (int j = 0; j < T; ++j)
                if (k.lane_ok[j]) Px::store_row(k.dst, k.xoff[j], v.v[j]);
            k.dst += k.pitch;
            k.row += 1;
            if (!STEADY && k.first_store < 0) k.first_store = it;
        }
    }
};

template <class Px, bool REV, int T, int NS, int LdsIdx, class S, class... Rest> struct Chain<Px, REV, T, NS, LdsIdx, S, Rest...> {
    static_assert(NS == 1 || SlotsOf<S>::value == 2, "every stage of a fork/join pipeline works on the pair (StSolo / StDup / StOn / StDelay / StMix)");
    typedef Tex<T * NS> V;
    static constexpr int K = BranchOf<S>::value > 0 ? BranchOf<S>::value : 0;      // the slot whose schedule this stage follows
    static constexpr bool kBranch = NS == 2 && BranchOf<S>::value >= 0;             // a stage of a branch: the other slot passes through
    static constexpr bool kFork = std::is_same<S, StDup>::value, kJoin = std::is_same<S, StMix>::value;
    static constexpr int RT = RowsAbove<S>::value, RB = RowsBelow<S>::value;
    typename S::template State<Px, T> st;
    // wave-uniform schedule
    int a;        // first input row
    int oa;       // first output row
    int flush;    // replications of the last input row (frame bottom edge)
    int tprev;    // iteration of the upstream stage's last emission (into this stage's slot)
    int cnt;      // input rows consumed
    int sa[NS], sb[NS], bb;      // the fork only: the rows each slot's stream starts and ends with, the fork's own last row
    Chain<Px, REV, T, NS, LdsIdx + S::LDS_ROWS, Rest...> next;

    // given the rows the LAST stage must emit, derive what each stage must emit/consume
    RF_DEV void plan_backward(int oa_last, int ob_last, int lo, int hi, Rows<NS>& in)
    {
        Rows<NS> need;
        next.plan_backward(oa_last, ob_last, lo, hi, need);
        in = need;
        if constexpr (kFork) {
            // one stream in, two out: it must carry every row either slot wants; each slot is handed its own rows only
#pragma unroll
            for (int k = 0; k < NS; ++k) { sa[k] = need.a[k]; sb[k] = need.b[k]; }
            a = oa = min(need.a[0], need.a[NS - 1]);
            bb = max(need.b[0], need.b[NS - 1]);
            flush = 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) { in.a[k] = a; in.b[k] = bb; }
        } else if constexpr (kJoin) {
            a = oa = need.a[0];
            flush = 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) { in.a[k] = need.a[0]; in.b[k] = need.b[0]; }
        } else {
            oa = need.a[K];
            a = max(lo, need.a[K] - RT);
            const int b = min(hi, need.b[K] + RB);
            flush = need.b[K] + RB - b;
            in.a[K] = a;
            in.b[K] = b;
        }
        cnt = 0;
    }
    RF_DEV int plan_forward(Ticks<NS> tp)
    {
        tprev = tp.t[K];
        if constexpr (kFork) {
            // row r of the fork's stream arrives (bb - r) iterations before its last one
            const int last = tp.t[0];
#pragma unroll
            for (int k = 0; k < NS; ++k) tp.t[k] = last - (bb - sb[k]);
        } else if constexpr (kJoin) {
#pragma unroll
            for (int k = 0; k < NS; ++k) tp.t[k] = tp.t[0];
        } else {
            tp.t[K] += flush;
        }
        return next.plan_forward(tp);
    }
    RF_DEV f4* lds_of(const Lane<T>& L) const { return L.lds + LdsIdx * 64 * T; }
    // the source feeds slot 0 (the first stage of a pair pipeline is an StSolo or the StDup itself)
    RF_DEV static V widen(const Tex<T>& a)
    {
        V o = tex_zero<T * NS>();
#pragma unroll
        for (int j = 0; j < T; ++j) o.v[j] = a.v[j];
        return o;
    }
    // which slots carry a row after this stage: the fork hands row r (the one it has just consumed) to the slots that want it,
    // the join emits when it ran, every other stage answers for its own slot and leaves the other bit alone
    RF_DEV unsigned bits_after(unsigned hb, bool ran, bool has) const
    {
        if constexpr (kFork) {
            if (!ran) return 0u;
            const int r = a + cnt - 1;
            unsigned ob = 0u;
#pragma unroll
            for (int k = 0; k < NS; ++k) ob |= (r >= sa[k] && r <= sb[k]) ? (1u << k) : 0u;
            return ob;
        } else if constexpr (kJoin) {
            return has ? 1u : 0u;
        } else {
            return (hb & ~(1u << K)) | (has ? (1u << K) : 0u);
        }
    }
    // the slot a branch stage does not work on passes through, whether or not the stage had a row of its own this iteration
    RF_DEV static void pass_other(const V& v, V& out)
    {
        if constexpr (kBranch) put_slot<T>(out, 1 - K, take_slot<T>(v, 1 - K));
    }
    // A row (or a flush tick) enters this stage.  STEADY = every stage receives a real row,
    // is past its first row and emits: the schedule tests fold away at compile time.
    template <int MODE> RF_DEV void step(unsigned hb, const V& v, int it, const Lane<T>& L, Sink<T>& k, const ParamPack<S, Rest...>& P)
    {
        constexpr bool STEADY = MODE != 0;
        constexpr bool KEEP = MODE == 0 || MODE == 3;   // phases that may hold a stage's last real row
        // this stage's own input: its slot's bit (the join: both slots, which arrive together by construction)
        const bool has_prev = kJoin ? ((hb & ((1u << NS) - 1u)) == ((1u << NS) - 1u)) : (((hb >> K) & 1u) != 0u);
        bool has = false, ran = false;
        V out = tex_zero<T * NS>();
        if constexpr (STEADY) {
            S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), v, true, false, true, out);
            cnt += 1;
            has = ran = true;
        } else if constexpr (RT == 0 && RB == 0) {
            if (has_prev) {              // row-local stage: one row in, one row out, never flushed
                S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), v, true, cnt == 0, true, out);
                cnt += 1;
                has = ran = true;
            }
        } else {
            const bool flushing = !has_prev && it > tprev && it <= tprev + flush;
            if (has_prev || flushing) {
                has = (a + cnt - RB) >= oa;
                S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), v, has_prev, cnt == 0, has, out);
                cnt += 1;
                ran = true;
            }
        }
        pass_other(v, out);
        next.template step<MODE>(STEADY ? hb : bits_after(hb, ran, has), out, it, L, k, P.rest);
    }
    // first stage: the row comes from the source feed; once it is consumed its ring slot is
    // refilled and the NEXT row's values are fetched into registers
    // MODE 0: generic (schedule tests).  Modes 1-3 are branch-free: every stage takes a real row
    // and emits one.  1 = rows still being issued, stores of the last PF iterations not all
    // there yet (wait on the loads alone); 2 = the steady state; 3 = every row issued already
    // (the last PF source rows): nothing to issue, plain wait.
    template <int MODE, class Feed, class Src>
    RF_DEV void step_first(bool has0, Feed& feed, const Src& src, int it, const Lane<T>& L, Sink<T>& k, const ParamPack<S, Rest...>& P)
    {
        constexpr bool STEADY = MODE != 0;
        constexpr bool KEEP = MODE == 0 || MODE == 3;
        static_assert(!kBranch && !kJoin, "a pipeline starts with a stage of slot 0 or with the fork");
        bool has = false, ran = false;
        V out = tex_zero<T * NS>();
        if constexpr (STEADY) {
            if constexpr (std::is_same<Feed, OwnFeed<Px, T>>::value)
                S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), widen(feed.own()), true, false, true, out);
            else
                out = S::template from_taps<T>(P.p, feed);
            cnt += 1;
            has = ran = true;
            if constexpr (MODE == 3) {
                if (it + 1 < src.n0) {
                    wait_vmcnt<0>();
                    feed.fetch(src, it + 1, L);
                }
            } else {
                src.issue(it + Src::SLOTS);
                if constexpr (MODE == 1) wait_vmcnt<(Src::SLOTS - 1) * T>();
                else wait_vmcnt<(2 * Src::SLOTS - 2) * T>();
                feed.fetch(src, it + 1, L);
            }
        } else {
            if constexpr (RT == 0 && RB == 0) {
                if (has0) {
                    if constexpr (std::is_same<Feed, OwnFeed<Px, T>>::value)
                        S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), widen(feed.own()), true, cnt == 0, true, out);
                    else
                        out = S::template from_taps<T>(P.p, feed);
                    cnt += 1;
                    has = ran = true;
                }
            } else {
                const bool flushing = !has0 && it > tprev && it <= tprev + flush;
                if (has0 || flushing) {
                    has = (a + cnt - RB) >= oa;
                    S::template advance<Px, REV, T, KEEP>(P.p, st, L, lds_of(L), widen(feed.own()), has0, cnt == 0, has, out);
                    cnt += 1;
                    ran = true;
                }
            }
            if (has0) {
                if (it + Src::SLOTS < src.n0) src.issue(it + Src::SLOTS);
                if (it + 1 < src.n0) {
                    src.wait_row(it + 1, k);
                    feed.fetch(src, it + 1, L);
                }
            }
        }
        next.template step<MODE>(STEADY ? ~0u : bits_after(has0 ? 1u : 0u, ran, has), out, it, L, k, P.rest);
    }
};

struct alignas(8) StreamHdr {
    const char* src;
    size_t src_pitch;
    char* dst;
    size_t dst_pitch;
    int W, row_lo, row_hi, y0, y1, rows_per_chunk, n_strips;
    int n_work;   // workgroups with work = strip groups x chunks (the grid is padded to a multiple of 8)
    int alternate;   // odd chunks walk bottom-up (halo rows shared through L2)
    // a second range of output rows [yb0, yb1) served by the chunks from index `chunks_a` on (Geom::yb0: the two boundary
    // slivers of a row strip as one launch); a launch with one range has chunks_a = its chunk count and an empty second range
    int chunks_a, yb0, yb1;
    int reserved;
};
template <class... S> struct StreamArgs : StreamHdr {
    ParamPack<S...> params;      // at offset sizeof(StreamHdr) = 88
};
static_assert(sizeof(StreamHdr) == 88, "the host assembles kernel arguments as bytes: header, then the parameter slots");

#ifndef RF_WAVES_PER_BLOCK
#define RF_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = RF_WAVES_PER_BLOCK;

// One wave's walk over rows [y0, y1) of its strip.  REV = bottom-up: rows are addressed with
// negated pitches and mirrored bounds, so the schedule code sees an ordinary top-down walk.
template <class Px, int PF, int T, bool REV, class... S>
RF_DEV void stream_wave(const StreamArgs<S...>& A, const Lane<T>& L, int wave, char* ring_wave, unsigned ring_lds, int y0, int y1)
{
    constexpr int RH = SumRH<S...>::value;
    typedef Source<Px, PF, T> Src;
    typedef typename FirstOf<S...>::type::template Feed<Px, T> Feed;
    (void)wave;

    // the walk's own row coordinate v: v = y top-down, v = -y bottom-up
    const int v0 = REV ? -(y1 - 1) : y0, v1 = REV ? -y0 + 1 : y1;
    const int lo = REV ? -A.row_hi : A.row_lo, hi = REV ? -A.row_lo : A.row_hi;

    Sink<T> k;
    k.dst = A.dst;
    k.pitch = REV ? -(ptrdiff_t)A.dst_pitch : (ptrdiff_t)A.dst_pitch;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        k.xoff[j] = (unsigned)min(max(L.col(j), 0), A.W - 1) * (unsigned)Px::BPP;
        k.lane_ok[j] = (L.pos(j) >= RH) && (L.pos(j) < 64 * T - RH) && (L.col(j) < A.W);
    }
    k.row = v0;
    k.dst = A.dst + (ptrdiff_t)v0 * k.pitch;
    k.first_store = -1;

    Chain<Px, REV, T, MaxSlots<S...>::value, 0, S...> chain;
    Src src;
    constexpr int kSlots = MaxSlots<S...>::value;
    Rows<kSlots> in;
    chain.plan_backward(v0, v1 - 1, lo, hi, in);
    src.a0 = in.a[0];
    src.n0 = in.b[0] - in.a[0] + 1;                  // source rows
    Ticks<kSlots> last;
#pragma unroll
    for (int q = 0; q < kSlots; ++q) last.t[q] = src.n0 - 1;
    const int total = chain.plan_forward(last) + 1;

    // source: rows a0..b0, column clamp(x)
    src.src = A.src;
#pragma unroll
    for (int j = 0; j < T; ++j) src.xoff[j] = k.xoff[j];
    src.pitch = REV ? -(ptrdiff_t)A.src_pitch : (ptrdiff_t)A.src_pitch;
    src.ring = ring_wave;
    src.lds_base = ring_lds;
    src.prologue();
    Feed feed;
    src.wait_row(0, k);
    feed.fetch(src, 0, L);

    // Phases: the generic loop (per-stage schedule tests) until the pipeline has emitted its
    // first row -- from then on every stage takes a real row and emits one for as long as
    // source rows arrive, and the branch-free modes run: 1 while the stores of the last PF
    // iterations are not all there yet, 2 the steady state, 3 the last PF source rows (nothing
    // left to issue) -- and the generic loop again for the bottom-edge flush.
    int it = 0;
    while (it < total && k.first_store < 0) {
        chain.template step_first<0>(it < src.n0, feed, src, it, L, k, A.params);
        ++it;
    }
    if (k.first_store >= 0) {
        const int steady_end = src.n0 - PF;                      // iterations with a row left to issue
        const int warm_end = min(k.first_store + PF, steady_end);
        for (; it < warm_end; ++it) chain.template step_first<1>(true, feed, src, it, L, k, A.params);
        for (; it < steady_end; ++it) chain.template step_first<2>(true, feed, src, it, L, k, A.params);
        for (; it < src.n0; ++it) chain.template step_first<3>(true, feed, src, it, L, k, A.params);
    }
    for (; it < total; ++it) chain.template step_first<0>(it < src.n0, feed, src, it, L, k, A.params);
}

template <class Px, int PF, int T, class... S>
__global__ __launch_bounds__(64 * kWavesPerBlock, T > 1 ? 2 : 1) void stream_kernel(const StreamArgs<S...> A)
{
    constexpr int RH = SumRH<S...>::value;
    constexpr int VALID = 64 * T - 2 * RH;
    constexpr int LDSR = SumLDS<S...>::value;
    typedef Source<Px, PF, T> Src;
    __shared__ f4 smem[kWavesPerBlock][(LDSR > 0 ? LDSR : 1) * 64 * T];
    __shared__ __attribute__((aligned(16))) char ring[kWavesPerBlock][Src::SLOTS * Src::SLOT_BYTES];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware block order (guide T1): blocks are dealt round-robin over the 8 XCDs, so block
    // b and b+8 share an L2.  Give each XCD a CONTIGUOUS range of work items (strip groups
    // fastest, then chunks): workgroups that share halo columns or halo rows then share an L2.
    // Speed only -- any placement gives the same result.
    // (Measured and NOT kept, profiles/r03_work_order_probe.txt, r03_stacked_chunks_probe.txt: workgroups in plain dispatch order,
    // chunks fastest instead of strip groups, and "stacked" workgroups whose four waves walk four vertically adjacent chunks of
    // one strip in alternating directions so that the seams' halo rows are shared inside the workgroup -- none was faster.)
    const int gx = (A.n_strips + kWavesPerBlock - 1) / kWavesPerBlock;
    const int per_xcd = (int)gridDim.x >> 3;
    const int q = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (q >= A.n_work) return;
    const int strip = (q % gx) * kWavesPerBlock + wave;
    if (strip >= A.n_strips) return;                 // wave-uniform; no barriers below
    const int chunk = q / gx;
    const bool second = chunk >= A.chunks_a;          // wave-uniform
    const int y0 = second ? A.yb0 + (chunk - A.chunks_a) * A.rows_per_chunk : A.y0 + chunk * A.rows_per_chunk;
    const int y1 = min(y0 + A.rows_per_chunk, second ? A.yb1 : A.y1);
    if (y0 >= y1) return;

    Lane<T> L;
    L.lane = (int)(threadIdx.x & 63);
    L.x0 = strip * VALID - RH;
    // T > 1: every one of the T stores of a row must have an active lane (the counted vmcnt waits
    // assume T stores are really issued; hipcc branches around a store whose exec mask is empty).
    // The last strip is therefore moved left until it ends at the frame edge -- it recomputes a
    // few columns of its neighbour and writes the same values (the host launches T > 1 only when
    // W >= 64 T and the launch is not in place).
    if constexpr (T > 1) {
        if (L.x0 + 64 * T - RH > A.W) L.x0 = A.W - 64 * T + RH;
    }
    L.W = A.W;
    L.lds = smem[wave];
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)(&ring[0][0]) + (unsigned)wave * (unsigned)(Src::SLOTS * Src::SLOT_BYTES));

    // Odd chunks walk bottom-up: a chunk and its neighbour then read the halo rows they
    // share at the same moment (both at their start, or both at their end), so the second
    // read is served by the XCD's L2 instead of the fabric.  Stencil-free pipelines have no
    // halo and always walk top-down.
    constexpr bool kHasHalo = SumRV<S...>::value > 0;
    if (kHasHalo && A.alternate && (chunk & 1))
        stream_wave<Px, PF, T, true, S...>(A, L, wave, ring[wave], ring_lds, y0, y1);
    else
        stream_wave<Px, PF, T, false, S...>(A, L, wave, ring[wave], ring_lds, y0, y1);
}

}  // namespace rf
