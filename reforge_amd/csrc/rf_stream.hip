// rf_stream.hip -- the streaming stage-pipeline kernel for gfx950 (CDNA4, wave64): every stencil / point node of the reforge
// render-graph path, alone or fused into chains.  It replaces shaders/*.comp + the per-node
// vkCmdDispatch of src/vulkan/command.rs:166-242.
//
// Design: the stencil/point nodes are HBM-bound (32 B/px for an
// rgba32f node), so the kernel is built to move every input row across the fabric
// once and keep everything else on chip:
//
//   * WAVE-AUTONOMOUS STREAMING PIPELINE.  Each 64-lane wave owns a column strip
//     64 pixels wide (one 16-byte texel per lane: one 1 KiB fully coalesced row segment)
//     and walks down (or up) a chunk of rows.  A node is a short list of row stages
//     (horizontal taps, vertical taps, point op, 3x3 cross); a fused chain of nodes is simply
//     a longer list.  Vertical taps keep a rolling window of rows in VGPRs; horizontal taps
//     exchange the current row between lanes through a wave-private 1 KiB LDS row (halo = the
//     wave's own edge lanes), so there is no workgroup barrier anywhere in the kernel -- LDS
//     operations of one wave execute in order.
//   * Input rows arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR destination) into a
//     wave-private ring of PF row slots, PF rows ahead, and are waited for with a COUNTED
//     s_waitcnt vmcnt (struct Source).  tests/test_isa_invariants.py checks the generated code
//     for the one-DMA-one-store-per-row contract that count rests on.
//   * Clamp-to-edge is applied where a stage READS (row index and lane index are
//     clamped to the image), which is what makes chained stages bit-identical to
//     running the nodes one full-frame pass at a time.
//
// Numerics: every multiply-add is an explicit fmaf in the oracle's tap order (packed two at a
// time, each half still a single-rounding fma); this file is compiled with -ffp-contract=off,
// so results are bit-identical to oracle/rf_oracle.c for finite inputs.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "rf_stream_dev.h"
#include "rf_jit.h"
#include "rf_user.h"

namespace rf {

// ---------------------------------------------------------------------------------
// Host side: op list -> stage list -> kernel instantiation
// ---------------------------------------------------------------------------------
static int device_cus()
{
    static int cus_of[64] = {};                      // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus_of[dev] == 0) {
        int cus = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        cus_of[dev] = cus > 0 ? cus : 256;
    }
    return cus_of[dev];
}

static int choose_rows_per_chunk(int rows, int strip_groups, int slots, int halo_rows, int bpp, const StreamTuning& tune, bool top_down)
{
    if (tune.rows_per_chunk > 0) return tune.rows_per_chunk;
    // A wave walks its chunk row by row (0.2-0.5 us per row), so a small frame is bound by the
    // LENGTH of the walks, not by HBM: take the shortest chunks whose workgroups are all resident
    // at once (`slots` = CUs x workgroups per CU of this kernel: one round, no tail).  A large
    // frame does not fit one round at any sensible height; there the chunk height is capped,
    // because more and shorter walks stream better, down to where the halo rows a chunk
    // recomputes (2*halo_rows) start to cost more than they hide.  Fitted to scripts/rpc_sweep.py
    // on MI355X, 720p..8K, both formats (best chunk heights: passthrough/sharpen 8-12 at every
    // size for rgba32f; 3-stage chain 12/16/24/24/24; gaussian9 12/24/24/32/32; 5-stage chain
    // 12/24/32/64/96 -- e.g. the 5-stage chain at 1080p 34 us with 24-row chunks, 66 us with 112):
    const int h = halo_rows;
    const bool narrow = bpp == 4;                    // rgba8: bound by its arithmetic, every halo row a chunk recomputes costs in full (7680x4320:
                                                     // gaussian9 40-row chunks 101 us, 85-row 95, 128-row 96; 3-stage chain 28-row 121, 96-row 113)
    // shortest chunks with every workgroup resident: at most slots / strip_groups chunks (a
    // 31-tap gaussian at 4K, one workgroup per CU: 9 chunks x 29 groups = 261 workgroups on 256
    // slots ran two rounds, 457 us; 8 chunks = 232 workgroups, 271 us)
    const int max_chunks = slots / strip_groups > 1 ? slots / strip_groups : 1;
    const long fit = (rows + max_chunks - 1) / max_chunks;
    const int lo = h > 4 ? 2 * h : 8;
    // all-top-down walks share no halo rows through L2: each chunk fetches its own, so they are taller (16384^2 5-stage chain,
    // two texels per lane: 112-row chunks 1.87 ms, 256-row 1.81, 384-row 2.01)
    // (round 3, with the non-temporal result stores: gaussian9 at 8K -- h = 4 -- takes 24-row chunks, never worse and 1.5-3 % better than
    // 30 / 32 / 36 on four boxes, profiles/r03_rpc_rounds_ab.txt; the 3-stage chain -- h = 3 -- stays at 8 h = 24: 32-row chunks +2 %)
    const int hi = narrow ? (h == 0 ? 16 : (24 * h > 32 ? 24 * h : 32)) : (h == 4 ? 24 : h < 4 ? (8 * h > 8 ? 8 * h : 8) : (top_down ? 36 * h : 16 * h));   // 5-stage chain (h = 7) at 16384^2: 128-row 2.03 ms, 96-row 2.08, 84-row 2.12
    // a frame that fits ONE round at up to twice the cap keeps the single round (4K 3-stage chain:
    // 36-row chunks = 1020 workgroups on 1024 slots, 194.6k Mpx/s; 24-row 191k; 44-row 180k)
    int rpc = (int)(fit < lo ? lo : (fit > 2 * hi ? hi : fit));
    rpc = (rpc + 3) & ~3;
    // a few rounds: use the rounds in full -- as many chunks as that number of rounds holds.  (16384^2 5-stage chain, two
    // texels per lane, 512 slots, 36 strip groups: 256-row chunks = 2304 workgroups = 4.5 rounds, 1.85-1.90 ms; 71 chunks of
    // 231 rows = 4.99 rounds, 1.79 ms; 72 chunks = 5.06 rounds, 1.90 ms -- scripts/walk_probe.py, the cliff is that sharp)
    {
        const int chunks0 = (rows + rpc - 1) / rpc;
        const long rounds = ((long)chunks0 * strip_groups + slots - 1) / slots;
        if (rounds >= 2 && rounds <= (h <= 4 ? 5 : 8)) {      // light pipelines past five rounds: the tail of a round no longer shows (gaussian9 8K: 22-row = 24-row)
            const long chunks = rounds * slots / strip_groups;
            if (chunks > chunks0) {
                int r2 = (int)((rows + chunks - 1) / chunks);
                if (r2 >= lo) rpc = r2;
            }
        }
    }
    // ONE round: what counts then is the busiest CU.  The dispatcher deals workgroups evenly, so a launch of k x CUs workgroups
    // (just below) finishes together, while k.5 x CUs leaves half the CUs a workgroup longer than the rest.  Among the chunk
    // heights around the one chosen above take the one with the least rows on the busiest CU (4K 3-stage rgba32f chain, 17 strip
    // groups, 5 workgroups per CU: 32-row chunks = 1156 workgroups = 4.5 per CU, 42.5 us; 29 rows = 1275 = 4.98, 41.7; 36 rows =
    // 1020 = 3.98, 41.7; 34 rows = 4.25 per CU, 43.4; 44 rows = 3.3, 44.6 -- scripts/walk_probe.py, repeatable to 0.1 us)
    {
        const int cus = device_cus();
        const long wgs0 = (long)((rows + rpc - 1) / rpc) * strip_groups;
        if (wgs0 <= slots && wgs0 > cus && rpc < rows) {
            auto busiest = [&](int r) {
                const long wgs = (long)((rows + r - 1) / r) * strip_groups;
                return wgs > slots ? 1L << 40 : ((wgs + cus - 1) / cus) * (long)(r + 2 * h);
            };
            int best = rpc;
            long best_cost = busiest(rpc);
            const int from = rpc - rpc / 6 > lo ? rpc - rpc / 6 : lo, to = rpc + rpc / 4;
            for (int r = from; r <= to && r <= rows; ++r) {
                const long c = busiest(r);
                if (c < best_cost) { best_cost = c; best = r; }
            }
            rpc = best;
        }
    }
    if (rpc > rows) rpc = rows;
    if (rpc < 1) rpc = 1;
    return rpc;
}

// ---------------------------------------------------------------------------------
// Stage lists at run time.  A launch is described by the list of row stages its nodes contribute
// (ops_to_stages): that list selects the kernel -- from the ahead-of-time CATALOGUE when it holds
// the instantiation, otherwise compiled at graph creation (rf_jit.cpp) -- and lays out the
// parameter block, which the host assembles as bytes (ParamPack in rf_stream_dev.h).
// ---------------------------------------------------------------------------------
static std::string stage_letter(int kind, int r, int user = -1)
{
    if (kind == ST_USER) { const UserStage* u = user_stage_by_id(user); return "U" + (u ? u->ident : std::string("?")); }
    switch (kind) {
        case ST_HTAP: return "H" + std::to_string(r);
        case ST_VTAP: return "V" + std::to_string(r);
        case ST_GRADE: return "G";
        case ST_CROSS3: return "C";
        case ST_DUP: return "D";
        case ST_MIX: return "M";
        case ST_DELAY: return "L" + std::to_string(r);
        default: return "E";
    }
}
static std::string stage_type(int kind, int r, int user = -1)
{
    if (kind == ST_USER) { const UserStage* u = user_stage_by_id(user); return "rf::StUser<rfuser::" + (u ? u->ident + "::" + u->row_stage : std::string("missing::Stage")) + ">"; }
    switch (kind) {
        case ST_HTAP: return "rf::StHTap<" + std::to_string(r) + ">";
        case ST_VTAP: return "rf::StVTap<" + std::to_string(r) + ">";
        case ST_GRADE: return "rf::StGrade";
        case ST_CROSS3: return "rf::StCross3";
        case ST_DUP: return "rf::StDup";
        case ST_MIX: return "rf::StMix";
        default: return "rf::StNodeEnd";
    }
}

std::string StageList::key() const
{
    static const char* pre[] = {"", "s", "a", "b"};
    std::string k;
    for (int i = 0; i < n; ++i) k += pre[st[i].slot] + stage_letter(st[i].kind, st[i].r, st[i].user) + ' ';
    return k;
}

std::string StageList::type_list() const
{
    std::string t;
    for (int i = 0; i < n; ++i) {
        const std::string base = stage_type(st[i].kind, st[i].r, st[i].user);
        if (st[i].kind == ST_DELAY) t += std::string("rf::StDelay<") + (st[i].slot == SLOT_ON0 ? "0" : "1") + ", " + std::to_string(st[i].r) + ">";
        else if (st[i].kind == ST_DUP || st[i].kind == ST_MIX || st[i].slot == SLOT_PLAIN) t += base;
        else if (st[i].slot == SLOT_SOLO) t += "rf::StSolo<" + base + ">";
        else t += std::string("rf::StOn<") + (st[i].slot == SLOT_ON0 ? "0" : "1") + ", " + base + ">";
        if (i + 1 < n) t += ", ";
    }
    return t;
}

static int stage_rv(int kind, int r) { return kind == ST_VTAP || kind == ST_USER ? r : (kind == ST_CROSS3 ? 1 : 0); }
static int stage_rh(int kind, int r) { return (kind == ST_HTAP || kind == ST_USER) ? r : (kind == ST_CROSS3 ? 1 : 0); }
// reach of a list (SumRH / SumRV of rf_stream_dev.h): the stages outside the branches of a fork/join add up, of the two branches --
// which run side by side -- the larger counts
static int reach(const StageList& sl, int (*radius)(int, int))
{
    int solo = 0, on[2] = {0, 0};
    for (int i = 0; i < sl.n; ++i) {
        const int v = radius(sl.st[i].kind, sl.st[i].r);
        if (sl.st[i].slot == SLOT_ON0 || sl.st[i].slot == SLOT_ON1) on[sl.st[i].slot == SLOT_ON1] += v;
        else solo += v;
    }
    return solo + std::max(on[0], on[1]);
}
int StageList::sum_rh() const { return reach(*this, stage_rh); }
int StageList::sum_rv() const { return reach(*this, stage_rv); }
int StageList::max_rv() const { int v = 0; for (int i = 0; i < n; ++i) v = std::max(v, stage_rv(st[i].kind, st[i].r)); return v; }
int StageList::taps() const
{
    int v = 0;
    for (int i = 0; i < n; ++i) v += (st[i].kind == ST_HTAP || st[i].kind == ST_VTAP) ? 2 * st[i].r + 1 : (st[i].kind == ST_CROSS3 ? 5 : (st[i].kind == ST_GRADE ? 3 : (st[i].kind == ST_USER ? (st[i].r ? 9 : 2) : 0)));
    return v;
}
// registers the pipeline's loop-carried state takes (vertical windows / running sums, the sharpen rows, the first
// stage's prefetched taps, the delay lines of a fork/join) + a fixed allowance: what decides whether a list may be fused at all
int StageList::vgpr_estimate(int texels) const
{
    int v = 56 + (pair() ? 16 : 0);
    for (int i = 0; i < n; ++i) {
        if (st[i].kind == ST_VTAP) v += 4 * texels * (2 * st[i].r + 1);
        if (st[i].kind == ST_CROSS3) v += 16 * texels;
        if (st[i].kind == ST_USER && st[i].r > 0) v += 24 * texels + 12;      // two rows of (west, centre, east) + the neighbourhood handed to apply()
        if (st[i].kind == ST_HTAP) v += (i == 0 && texels == 1 && st[i].r <= 7 && !pair()) ? 4 * (2 * st[i].r + 1) : 0;
        if (st[i].kind == ST_DELAY) v += 4 * texels * st[i].r;     // the delay line of the shorter branch
    }
    int transient = 0;
    for (int i = 0; i < n; ++i)
        if (st[i].kind == ST_HTAP) transient = std::max(transient, 4 * texels * 2 * st[i].r);
    return v + transient / 2;
}

// ops -> stages.  A plain chain: the nodes' stages with a boundary (the store + load the unfused graph performs) between
// nodes.  A fork/join launch (ops carry slot 1 / 2 for the two branches, one OP_MIX is the join): nodes before the fork,
// StDup, branch 0, branch 1, a boundary per non-empty branch, StMix, nodes after the join -- every stage wrapped for the
// pair pipeline (rf_stream_dev.h, "Fork / join in ONE launch").
bool ops_to_stages(const Op* ops, int n, StageList& out)
{
    out.n = 0;
    int mix = -1;
    for (int i = 0; i < n; ++i)
        if (ops[i].kind == OP_MIX) { if (mix >= 0) return false; mix = i; }
    const bool pair = mix >= 0;
    auto push = [&](int kind, int r, int slot, int op) {
        if (out.n >= StageList::kMax) return false;
        out.st[out.n].kind = kind;
        out.st[out.n].r = r;
        out.st[out.n].slot = slot;
        out.st[out.n].op = op;
        out.st[out.n].user = (kind == ST_USER && op >= 0) ? ops[op].user_id : -1;
        ++out.n;
        return true;
    };
    auto last_is_end = [&](int slot) { return out.n > 0 && out.st[out.n - 1].kind == ST_NODE_END && out.st[out.n - 1].slot == slot; };
    // the stages of nodes [from, to) whose op.slot == want, wrapped as `slot`; `started`: something of this run came before (a
    // boundary goes between nodes; a LEADING passthrough is the load itself, no stage at all)
    auto run = [&](int from, int to, int want, int slot) {
        bool started = false;
        for (int i = from; i < to; ++i) {
            if (ops[i].slot != want || ops[i].kind == OP_MIX) continue;
            if (started && !last_is_end(slot))
                if (!push(ST_NODE_END, 0, slot, -1)) return false;
            switch (ops[i].kind) {
                case OP_PASSTHROUGH: break;
                case OP_GAUSSIAN:
                    if (ops[i].radius < 0 || ops[i].radius > kMaxRadius) return false;
                    if (!push(ST_HTAP, ops[i].radius, slot, i) || !push(ST_VTAP, ops[i].radius, slot, i)) return false;
                    started = true;
                    break;
                case OP_GRADE: if (!push(ST_GRADE, 0, slot, i)) return false; started = true; break;
                case OP_SHARPEN: if (!push(ST_CROSS3, 0, slot, i)) return false; started = true; break;
                case OP_USER: if (ops[i].user_id < 0 || !push(ST_USER, ops[i].radius, slot, i)) return false; started = true; break;
                default: return false;     // conv2d: a kernel of its own
            }
        }
        return true;
    };
    if (!pair) {
        for (int i = 0; i < n; ++i)
            if (ops[i].slot != 0) return false;
        if (!run(0, n, 0, SLOT_PLAIN)) return false;
        if (out.n == 0) return push(ST_NODE_END, 0, SLOT_PLAIN, -1);      // a chain of passthroughs: the copy kernel
        if (out.st[out.n - 1].kind == ST_NODE_END && out.n > 1) --out.n;   // a trailing boundary is the final store itself
        return true;
    }
    // ops order of a fork/join launch: [before the fork (slot 0)] [branch 0 (slot 1)] [branch 1 (slot 2)] [OP_MIX] [after the join (slot 0)]
    int first_branch = mix;
    for (int i = 0; i < mix; ++i)
        if (ops[i].slot != 0) { first_branch = i; break; }
    for (int i = first_branch; i < mix; ++i)
        if (ops[i].slot == 0) return false;
    for (int i = mix + 1; i < n; ++i)
        if (ops[i].slot != 0) return false;
    if (!run(0, first_branch, 0, SLOT_SOLO)) return false;
    if (out.n > 0 && !last_is_end(SLOT_SOLO) && !push(ST_NODE_END, 0, SLOT_SOLO, -1)) return false;   // the forked image is stored and loaded
    if (!push(ST_DUP, 0, SLOT_SOLO, -1)) return false;
    int branch_rv[2] = {0, 0};
    for (int i = first_branch; i < mix; ++i)
        if (ops[i].slot == 1 || ops[i].slot == 2) branch_rv[ops[i].slot - 1] += (ops[i].kind == OP_GAUSSIAN || ops[i].kind == OP_USER) ? ops[i].radius : (ops[i].kind == OP_SHARPEN ? 1 : 0);
    for (int b = 0; b < 2; ++b) {
        const int before = out.n, slot = b == 0 ? SLOT_ON0 : SLOT_ON1;
        if (!run(first_branch, mix, b + 1, slot)) return false;
        if (out.n > before && !last_is_end(slot) && !push(ST_NODE_END, 0, slot, -1)) return false;    // the branch result is stored and loaded by the join
        // the branches run side by side: the one of the smaller vertical radius ends in a delay line, so that both reach the join
        // with the same frame row (rf_stream_dev.h, StDelay)
        if (branch_rv[b] < branch_rv[1 - b] && !push(ST_DELAY, branch_rv[1 - b] - branch_rv[b], slot, -1)) return false;
    }
    if (!push(ST_MIX, 0, SLOT_SOLO, mix)) return false;
    const int after = out.n;
    bool any_after = false;
    for (int i = mix + 1; i < n; ++i) any_after = any_after || ops[i].kind != OP_PASSTHROUGH;
    if (any_after) {
        if (!push(ST_NODE_END, 0, SLOT_SOLO, -1)) return false;
        if (!run(mix + 1, n, 0, SLOT_SOLO)) return false;
        if (out.n > after && last_is_end(SLOT_SOLO)) --out.n;
    }
    return true;
}

// parameter block of a stage list from the ops of its nodes, as bytes (layout: ParamPack, rf_stream_dev.h)
static size_t param_bytes(const StageList& sl, const Op* ops, int n_ops, unsigned char* buf, size_t cap)
{
    size_t off = 0;
    std::memset(buf, 0, cap);
    for (int i = 0; i < sl.n; ++i) {
        const int kind = sl.st[i].kind, r = sl.st[i].r;
        size_t size = 8;
        if (kind == ST_HTAP || kind == ST_VTAP) size = 8 * (size_t)(r + 1);
        else if (kind == ST_GRADE || kind == ST_CROSS3) size = 16;
        else if (kind == ST_USER) { const UserStage* u = user_stage_by_id(sl.st[i].user); if (!u) return 0; size = (size_t)((u->params_size + 7) / 8 * 8); }
        if (off + size + 8 > cap) return 0;
        unsigned char* p = buf + off;
        const Op* op = (sl.st[i].op >= 0 && sl.st[i].op < n_ops) ? &ops[sl.st[i].op] : nullptr;
        if ((kind == ST_HTAP || kind == ST_VTAP) && op) {
            for (int k = 0; k <= r; ++k) {
                const float pr[2] = {op->w[k], op->w[k]};
                std::memcpy(p + 8 * k, pr, 8);
            }
        } else if (kind == ST_GRADE && op) {
            const float g[3] = {op->slope, op->offset, op->saturation};
            std::memcpy(p, g, 12);
        } else if (kind == ST_CROSS3 && op) {
            const float c[4] = {op->wc, op->wc, op->ws, op->ws};
            std::memcpy(p, c, 16);
        } else if (kind == ST_USER && op) {
            std::memcpy(p, op->user_params, std::min(size, sizeof(op->user_params)));
        } else if (kind == ST_MIX && op) {
            std::memcpy(p, &op->slope, 4);       // OP_MIX keeps its factor in `slope`
        }
        off += size;
    }
    return off + 8;     // the closing empty slot
}

// ---------------------------------------------------------------------------------
// Launch geometry shared by catalogue and run-time compiled kernels
// ---------------------------------------------------------------------------------
struct LaunchShape {
    StreamHdr hdr;
    unsigned grid;
};

static bool launch_shape(Image src, Image dst, const Geom& g, const StreamTuning& tune, int texels, int rh, int halo_rows, int bpp,
                         int resident, LaunchShape& out)
{
    const int VALID = 64 * texels - 2 * rh;
    StreamHdr& A = out.hdr;
    A.src = static_cast<const char*>(src.base);
    A.src_pitch = src.pitch;
    A.dst = static_cast<char*>(dst.base);
    A.dst_pitch = dst.pitch;
    A.W = g.W;
    A.row_lo = g.row_lo;
    A.row_hi = g.row_hi;
    A.y0 = g.y0;
    A.y1 = g.y1;
    A.n_strips = (g.W + VALID - 1) / VALID;
    const int rows_a = g.y1 - g.y0 > 0 ? g.y1 - g.y0 : 0, rows_b = g.yb1 - g.yb0 > 0 ? g.yb1 - g.yb0 : 0;
    const int rows = rows_a + rows_b;
    if (rows <= 0 || g.W <= 0) return false;
    const int groups = (A.n_strips + kWavesPerBlock - 1) / kWavesPerBlock;
    A.rows_per_chunk = choose_rows_per_chunk(rows_b > 0 ? std::max(rows_a, rows_b) : rows, groups, resident, halo_rows, bpp, tune, tune.walk == 2);
    A.chunks_a = (rows_a + A.rows_per_chunk - 1) / A.rows_per_chunk;
    A.yb0 = g.yb0;
    A.yb1 = g.yb0 + rows_b;
    const int chunks = A.chunks_a + (rows_b + A.rows_per_chunk - 1) / A.rows_per_chunk;
    A.alternate = tune.walk == 2 ? 0 : 1;
    A.reserved = 0;
    A.n_work = groups * chunks;
    out.grid = (unsigned)((A.n_work + 7) / 8 * 8);   // 1-D, a multiple of the 8 XCDs (see the kernel's block order)
    static const bool trace = std::getenv("RF_TRACE_SHAPE") != nullptr;      // diagnostics: the launch geometry, to stderr
    if (trace)
        std::fprintf(stderr, "rf shape: %dx%d rows, bpp %d, texels %d, walk %d: %d strips in %d groups, %d-row chunks, %d workgroups on %d resident (%d CUs)\n",
                     g.W, rows, bpp, texels, tune.walk, A.n_strips, groups, A.rows_per_chunk, A.n_work, resident, device_cus());
    return true;
}

// workgroups of this kernel the whole chip holds at once
template <class Px, int PF, int T, class... S> static int resident_workgroups()
{
    static int slots_of[64] = {};                    // per device: a process may hold contexts on several GPUs
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int& slots = slots_of[dev];
    if (slots == 0) {
        int per_cu = 0, cus = 256;
        const size_t dyn = std::getenv("RF_EXPERIMENT_DYN_LDS") ? (size_t)std::atoi(std::getenv("RF_EXPERIMENT_DYN_LDS")) : 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stream_kernel<Px, PF, T, S...>, 64 * kWavesPerBlock, dyn) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = per_cu * (cus > 0 ? cus : 256);
    }
    return slots;
}

// one catalogue entry: stream_kernel<Px, PF, T, S...> behind a type-erased launcher
template <class Px, int PF, int T, class... S>
static hipError_t launch_aot(Image src, Image dst, const Geom& g, const StreamTuning& tune, hipStream_t stream,
                             const unsigned char* pbytes, size_t psize, int halo_rows)
{
    constexpr int RH = SumRH<S...>::value;
    static_assert(64 * T - 2 * RH > 0, "horizontal halo too wide for the strip");
    static_assert(sizeof(StreamArgs<S...>) == sizeof(StreamHdr) + sizeof(ParamPack<S...>), "header, then the parameter slots");
    if (psize != sizeof(ParamPack<S...>)) return hipErrorInvalidValue;   // the byte layout and the template disagree: a bug, never launch
    LaunchShape sh;
    if (!launch_shape(src, dst, g, tune, T, RH, halo_rows, Px::BPP, resident_workgroups<Px, PF, T, S...>(), sh)) return hipSuccess;
    StreamArgs<S...> A;
    static_cast<StreamHdr&>(A) = sh.hdr;
    std::memcpy(&A.params, pbytes, psize);
    // EXPERIMENT (RF_EXPERIMENT_DYN_LDS = bytes of unused dynamic LDS per workgroup): lowers the occupancy of a launch without
    // touching its code -- how much of a heavy pipeline's time is its two waves per SIMD (scripts/walk_probe.py)
    static const unsigned dyn_lds = std::getenv("RF_EXPERIMENT_DYN_LDS") ? (unsigned)std::atoi(std::getenv("RF_EXPERIMENT_DYN_LDS")) : 0u;
    hipLaunchKernelGGL((stream_kernel<Px, PF, T, S...>), dim3(sh.grid), dim3(64 * kWavesPerBlock), dyn_lds, stream, A);
    return hipGetLastError();
}

typedef hipError_t (*AotFn)(Image, Image, const Geom&, const StreamTuning&, hipStream_t, const unsigned char*, size_t, int);
struct AotEntry {
    AotFn fn[3][3] = {};      // [format, or kPxF32Stream = rgba32f with non-temporal stores][texels per lane]
};

template <class S> struct StageKey;
template <int R> struct StageKey<StHTap<R>> { static std::string get() { return "H" + std::to_string(R) + " "; } };
template <int R> struct StageKey<StVTap<R>> { static std::string get() { return "V" + std::to_string(R) + " "; } };
template <> struct StageKey<StGrade> { static std::string get() { return "G "; } };
template <> struct StageKey<StCross3> { static std::string get() { return "C "; } };
template <> struct StageKey<StNodeEnd> { static std::string get() { return "E "; } };
template <> struct StageKey<StDup> { static std::string get() { return "sD "; } };
template <> struct StageKey<StMix> { static std::string get() { return "sM "; } };
template <class S> struct StageKey<StSolo<S>> { static std::string get() { return "s" + StageKey<S>::get(); } };
template <int K, class S> struct StageKey<StOn<K, S>> { static std::string get() { return (K == 0 ? "a" : "b") + StageKey<S>::get(); } };
template <int K, int D> struct StageKey<StDelay<K, D>> { static std::string get() { return std::string(K == 0 ? "a" : "b") + "L" + std::to_string(D) + " "; } };

#ifndef RF_PF_DEFAULT
#define RF_PF_DEFAULT 4
#endif
constexpr int PF_DEFAULT = RF_PF_DEFAULT;   // rows in flight per wave (8 was measured slower, 3/5/6 see DESIGN.md)
#ifndef RF_PF_T2
#define RF_PF_T2 4
#endif

constexpr int kHeavyTaps = 24;      // multiply-adds per texel from which a pipeline counts as issue-bound (walk policy below, stream_kernel_code)
// the same count the run-time StageList::taps() makes, at compile time: which catalogue entries need a non-temporal-store variant
template <class S> struct StageTaps { static constexpr int value = 0; };
template <int R> struct StageTaps<StHTap<R>> { static constexpr int value = 2 * R + 1; };
template <int R> struct StageTaps<StVTap<R>> { static constexpr int value = 2 * R + 1; };
template <> struct StageTaps<StGrade> { static constexpr int value = 3; };
template <> struct StageTaps<StCross3> { static constexpr int value = 5; };
template <class S> struct StageTaps<StSolo<S>> { static constexpr int value = StageTaps<S>::value; };
template <int K, class S> struct StageTaps<StOn<K, S>> { static constexpr int value = StageTaps<S>::value; };
template <class... S> struct SumTaps { static constexpr int value = 0; };
template <class S, class... Rest> struct SumTaps<S, Rest...> { static constexpr int value = StageTaps<S>::value + SumTaps<Rest...>::value; };

template <class... S> struct TL {};
template <class A, class B> struct Concat;
template <class... A, class... B> struct Concat<TL<A...>, TL<B...>> { typedef TL<A..., B...> type; };
template <class A, class B> struct Join { typedef typename Concat<typename Concat<A, TL<StNodeEnd>>::type, B>::type type; };

static std::map<std::string, AotEntry>& catalogue()
{
    static std::map<std::string, AotEntry> c;
    return c;
}

template <int PF, class... S> static void add_to_catalogue(TL<S...>)
{
    std::string key;
    const int dummy[] = {0, (key += StageKey<S>::get(), 0)...};
    (void)dummy;
    AotEntry& e = catalogue()[key];
    e.fn[kFmtRGBA8][1] = &launch_aot<PxU8, PF, 1, S...>;
    e.fn[kFmtRGBA32F][1] = &launch_aot<PxF32, PF, 1, S...>;
    // the non-temporal-store variant only where stream_kernel_code can ask for it (issue-bound lists never do: a third of the catalogue,
    // and the slowest kernels to build)
    constexpr bool kNt = SumTaps<S...>::value < kHeavyTaps;
    if constexpr (kNt) e.fn[kPxF32Stream][1] = &launch_aot<PxF32NT, PF, 1, S...>;
    // two texels per lane: only where the doubled state still fits 256 VGPRs
    if constexpr (SumRH<S...>::value <= 7 && MaxRV<S...>::value <= 4 && MaxSlots<S...>::value == 1) e.fn[kFmtRGBA8][2] = &launch_aot<PxU8, (PF > 4 ? 4 : PF), 2, S...>;
    if constexpr (SumRH<S...>::value <= 7 && MaxRV<S...>::value <= 4 && MaxSlots<S...>::value == 1) e.fn[kFmtRGBA32F][2] = &launch_aot<PxF32, (PF > RF_PF_T2 ? RF_PF_T2 : (PF == PF_DEFAULT ? RF_PF_T2 : PF)), 2, S...>;
    if constexpr (kNt && SumRH<S...>::value <= 7 && MaxRV<S...>::value <= 4 && MaxSlots<S...>::value == 1) e.fn[kPxF32Stream][2] = &launch_aot<PxF32NT, (PF > RF_PF_T2 ? RF_PF_T2 : (PF == PF_DEFAULT ? RF_PF_T2 : PF)), 2, S...>;
}

// The ahead-of-time catalogue: every node alone (gaussian radius 0..15), every ordered pair of
// {gaussian5, gaussian9, grade, sharpen}, the triples gaussian(5|9) -> grade -> sharpen and the whole 5-stage chain of
// the BASELINE configs.  Any other chain of fusable nodes is compiled when a graph that contains it is created
// (rf_jit.cpp); with RF_GRAPH_NO_JIT, or without libhiprtc, it is cut into the longest pieces found here.
template <int R> static void add_gaussians()
{
    if constexpr (R <= kMaxRadius) {
        add_to_catalogue<(R <= 4 ? PF_DEFAULT : 2)>(TL<StHTap<R>, StVTap<R>>{});
        add_gaussians<R + 1>();
    }
}

template <int CODE> struct NodeTL;
template <> struct NodeTL<0> { typedef TL<StHTap<2>, StVTap<2>> type; };
template <> struct NodeTL<1> { typedef TL<StHTap<4>, StVTap<4>> type; };
template <> struct NodeTL<2> { typedef TL<StGrade> type; };
template <> struct NodeTL<3> { typedef TL<StCross3> type; };

template <int A, int B> static void add_pairs()
{
    if constexpr (A < 4) {
        add_to_catalogue<PF_DEFAULT>(typename Join<typename NodeTL<A>::type, typename NodeTL<B>::type>::type{});
        if constexpr (B + 1 < 4) add_pairs<A, B + 1>();
        else add_pairs<A + 1, 0>();
    }
}

static const std::map<std::string, AotEntry>& built_catalogue()
{
    static bool done = false;
    if (!done) {
        done = true;
        add_to_catalogue<PF_DEFAULT>(TL<StNodeEnd>{});
        add_to_catalogue<PF_DEFAULT>(TL<StGrade>{});
        add_to_catalogue<PF_DEFAULT>(TL<StCross3>{});
        add_gaussians<0>();
        add_pairs<0, 0>();
        typedef typename Join<NodeTL<2>::type, NodeTL<3>::type>::type GradeSharp;
        add_to_catalogue<PF_DEFAULT>(typename Join<NodeTL<0>::type, GradeSharp>::type{});
        add_to_catalogue<PF_DEFAULT>(typename Join<NodeTL<1>::type, GradeSharp>::type{});
        typedef typename Join<NodeTL<0>::type, GradeSharp>::type Head;
        typedef typename Join<NodeTL<1>::type, NodeTL<2>::type>::type Tail;
        add_to_catalogue<PF_DEFAULT>(typename Join<Head, Tail>::type{});      // gaussian5 -> grade -> sharpen -> gaussian9 -> grade
        // the fork/join example of pipeline_graph.rs:462-468 (gaussian5 || sharpen -> combination) as ONE launch
        add_to_catalogue<PF_DEFAULT>(TL<StDup, StOn<0, StHTap<2>>, StOn<0, StVTap<2>>, StOn<0, StNodeEnd>, StOn<1, StCross3>, StOn<1, StNodeEnd>, StDelay<1, 1>, StMix>{});
    }
    return catalogue();
}

bool stream_in_catalogue(const StageList& sl) { return built_catalogue().count(sl.key()) != 0; }

// what a run-time compiled kernel may be asked to hold: the planner's admission rule for a chain outside the catalogue
bool stream_jit_admissible(const StageList& sl)
{
    return sl.n >= 1 && sl.sum_rh() <= 12 && sl.vgpr_estimate(1) <= 224;      // (up to 256 VGPRs = two waves per SIMD)
}

bool stream_supported(const Op* ops, int n, bool allow_jit)
{
    if (n <= 0 || n > kMaxFusedOps) return false;
    StageList sl;
    if (!ops_to_stages(ops, n, sl)) return false;
    if (stream_in_catalogue(sl)) return true;
    return (n >= 2 || sl.has_user()) && allow_jit && stream_jit_admissible(sl) && jit_available();      // (a user stage is never in the catalogue: compiled even alone)
}

int ops_radius(const Op* ops, int n)
{
    // a fused fork/join launch (ops of slot 1 / 2 = its two branches) runs the branches side by side: pre + max(a, b) + post
    int r[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        int v = 0;
        switch (ops[i].kind) {
            case OP_GAUSSIAN: v = ops[i].radius; break;
            case OP_SHARPEN: v = 1; break;
            case OP_USER: v = ops[i].radius; break;
            case OP_USERN: v = ops[i].radius; break;
            case OP_CONV2D: v = ops[i].radius; break;
            default: break;
        }
        r[ops[i].slot >= 0 && ops[i].slot <= 2 ? ops[i].slot : 0] += v;
    }
    return r[0] + std::max(r[1], r[2]);
}

// Walk policy of a launch (measured with scripts/walk_probe.py on MI355X):
//  * ISSUE-BOUND pipelines (>= kHeavyTaps multiply-adds per texel) walk every chunk TOP-DOWN and, on large
//    rgba32f frames, take TWO texels per lane: 128-wide strips halve the share of halo
//    lanes and the per-row scalar work, and top-down walks run the vertical taps in scatter form
//    (StVTap: no window shifts).  5-stage chain: 16384^2 2.04 -> 1.86 ms, 8K 0.260 -> 0.241 ms.
//  * everything else keeps one texel per lane and ALTERNATING walks: it is bound by the memory
//    path, more waves in flight and halo rows shared through L2 matter more (3-stage chain at 4K:
//    42.2 us alternating, 43.4 top-down, 43.7 with two texels).
// Two texels per lane are never used in place (the last strip overlaps its neighbour: see
// stream_kernel), for rgba8 (its lanes would issue four 256-B DMAs per row) or for narrow frames.
constexpr long kTopDownMinPixels = 48L << 20;       // light rgba32f pipelines walk top-down from here up
constexpr long kTwoTexelLightMinPixels = 200L << 20;   // ... and take two texels per lane from here up
constexpr long kTwoTexelMinPixels = 24L << 20;   // two texels per lane from 8K frames up (5-stage chain at 4K: 64.4 us with one, 68.3 with two)

// Non-temporal row stores (PxF32NT, rf_device.h) for the result of a frame: measured on MI355X, interleaved on one box
// (profiles/r03_store_nt_probe.txt, r03_nt_store_final_ab.txt): gaussian9 at 8K -1.5..-4 %, the fused 4K chain -1.5..-2 %, the 8K
// chain -0.5 %; the issue-bound 5-stage chain at 16384^2 +0.4 % (two texels per lane) / +2.5 % (one) -- so not for heavy pipelines.
int stream_kernel_code(int fmt, const StageList& sl, bool nt_store)
{
    return (fmt == kFmtRGBA32F && nt_store && sl.taps() < kHeavyTaps) ? kPxF32Stream : fmt;
}

static int choose_texels(int fmt, const StageList& sl, Image src, Image dst, const Geom& g, StreamTuning& t)
{
    const long px = (long)g.W * (long)((g.y1 - g.y0) + (g.yb1 > g.yb0 ? g.yb1 - g.yb0 : 0));
    const bool heavy = sl.taps() >= kHeavyTaps;
    // rgba8 pipelines are bound by vector issue whatever their length (conversion arithmetic): top-down as well
    // (4K 3-stage chain 40.1 -> 39.2 us, 8K gaussian9 117.7 -> 114.3 us)
    // ... and so do the light pipelines once the frame is far beyond the Infinity Cache (gaussian9: 7680x4320 alternating 184 us,
    // top-down 189-197; 8192^2 413 -> 397; 16384x4096 416 -> 387; 12288^2 868 -> 847.  3-stage chain 12288^2 915 -> 869)
    if (t.walk == 0) t.walk = (heavy || fmt == kFmtRGBA8 || px >= kTopDownMinPixels) ? 2 : 1;
    if (sl.sum_rh() > 7 || sl.max_rv() > 4 || sl.pair()) return 1;
    const bool can2 = src.base != dst.base && g.W >= 256;
    // two texels per lane: heavy pipelines from 8K frames up, every rgba32f pipeline on the very largest frames (16384^2:
    // gaussian9 1.76 -> 1.53 ms, 3-stage chain 1.72 -> 1.61, passthrough 1.52 -> 1.46; 12288^2: no gain yet)
    const bool two = t.texels_per_lane == 2 ||
                     (t.texels_per_lane == 0 && fmt == kFmtRGBA32F && t.walk == 2 && px >= (heavy ? kTwoTexelMinPixels : kTwoTexelLightMinPixels));
    return (two && can2) ? 2 : 1;
}

// texels per lane a graph of this size would use for this list (so that rf_graph_create can compile that variant too)
int stream_texels_for(int fmt, const Op* ops, int n, int W, int rows, const StreamTuning& tune)
{
    StageList sl;
    if (!ops_to_stages(ops, n, sl)) return 1;
    StreamTuning t = tune;
    Geom g;
    g.W = W;
    g.y0 = 0;
    g.y1 = rows;
    Image a{(void*)16, 0}, b{(void*)32, 0};
    return choose_texels(fmt, sl, a, b, g, t);
}

static hipError_t launch_stages(int fmt, const StageList& sl, const Op* ops, int n, Image src, Image dst, const Geom& g,
                                const StreamTuning& tune, hipStream_t stream)
{
    if ((g.y1 - g.y0 <= 0 && g.yb1 - g.yb0 <= 0) || g.W <= 0) return hipSuccess;
    StreamTuning t = tune;
    const int texels = choose_texels(fmt, sl, src, dst, g, t);
    unsigned char pbytes[kMaxParamBytes];
    const size_t psize = param_bytes(sl, ops, n, pbytes, sizeof(pbytes));
    if (psize == 0) return hipErrorInvalidValue;
    const int halo = sl.sum_rv();
    const int kc = stream_kernel_code(fmt, sl, g.nt_store);      // which kernel: the format, or rgba32f with non-temporal stores
    auto it = built_catalogue().find(sl.key());
    if (it != built_catalogue().end()) {
        AotFn fn = it->second.fn[kc][texels];
        if (!fn) fn = it->second.fn[kc][1];
        if (!fn) fn = it->second.fn[fmt][texels] ? it->second.fn[fmt][texels] : it->second.fn[fmt][1];      // (no such variant: the format's own kernel)
        return fn(src, dst, g, t, stream, pbytes, psize, halo);
    }
    // compiled at graph creation (rf_graph_create -> stream_prepare); never compiled here, on the frame path
    const JitKernel* k = jit_lookup(kc, PF_DEFAULT, texels, sl);
    if (!k && texels != 1) k = jit_lookup(kc, PF_DEFAULT, 1, sl);
    if (!k) return hipErrorInvalidDeviceFunction;
    LaunchShape sh;
    const int bpp = fmt == kFmtRGBA8 ? 4 : 16;
    if (!launch_shape(src, dst, g, t, k->texels, sl.sum_rh(), halo, bpp, k->resident_workgroups, sh)) return hipSuccess;
    unsigned char args[sizeof(StreamHdr) + kMaxParamBytes];
    std::memcpy(args, &sh.hdr, sizeof(StreamHdr));
    std::memcpy(args + sizeof(StreamHdr), pbytes, psize);
    return jit_launch(*k, sh.grid, 64 * kWavesPerBlock, args, sizeof(StreamHdr) + psize, stream);
}

// rf_graph_create: make sure the kernel of this fused launch exists (compile it if the catalogue lacks it)
bool stream_prepare(int fmt, const Op* ops, int n, int W, int rows, const StreamTuning& tune, bool nt_store, std::string& err, std::string* note)
{
    StageList sl;
    if (!ops_to_stages(ops, n, sl)) { err = "not a streaming launch"; return false; }
    if (stream_in_catalogue(sl)) return true;
    const int real_fmt = fmt;
    fmt = stream_kernel_code(fmt, sl, nt_store);      // the variant this launch will ask for (launch_stages)
    if (!jit_compile(fmt, PF_DEFAULT, 1, sl, kWavesPerBlock, err)) return false;
    if (const JitKernel* k = jit_lookup(fmt, PF_DEFAULT, 1, sl)) {
        // The admission rule is an estimate; a FUSED chain that spills after all is not worth its launch: the caller plans it again
        // in pieces.  A launch that is ONE node cannot be cut further -- a user stage whose apply() needs scratch (a local array
        // indexed at run time, a heavy body): the reference runs every shader that compiles (shader.rs:29-93), so it is kept and the
        // spill reported (`note`).  Scratch accesses are the compiler's own vector-memory instructions: it waits for them itself
        // and the counted waits only become stricter by them -- slower, never wrong.
        if (k->scratch_bytes > 0) {
            const std::string what = "the compiled chain " + sl.key() + "spills " + std::to_string(k->scratch_bytes) + " bytes per lane";
            if (n > 1) { err = what; return false; }
            if (note) *note = what;
        }
    }
    if (stream_texels_for(real_fmt, ops, n, W, rows, tune) == 2) {
        std::string e2;
        // optional variant: the one-texel kernel serves if it fails to compile -- or if it spills (its state is twice the
        // one-texel kernel's under the same 256-VGPR bound), which the ahead-of-time kernels are checked for by the ISA test
        if (jit_compile(fmt, PF_DEFAULT, 2, sl, kWavesPerBlock, e2)) {
            const JitKernel* k2 = jit_lookup(fmt, PF_DEFAULT, 2, sl);
            if (k2 && k2->scratch_bytes > 0) jit_forget(fmt, PF_DEFAULT, 2, sl);
        }
    }
    return true;
}

template <class Px>
static hipError_t launch_ops_px(int fmt, const Op* ops, int n, Image src, Image dst, const Geom& g, const StreamTuning& tune,
                                hipStream_t stream)
{
    if (n == 1 && ops[0].kind == OP_CONV2D) return launch_conv2d(fmt, ops[0], src, dst, g, tune, stream);
    StageList sl;
    if (!ops_to_stages(ops, n, sl)) return hipErrorInvalidValue;
    return launch_stages(fmt, sl, ops, n, src, dst, g, tune, stream);
}

hipError_t launch_ops(int fmt, const Op* ops, int n, Image src, Image dst, const Geom& g, const StreamTuning& tune,
                      hipStream_t stream)
{
    if (fmt == kFmtRGBA8) return launch_ops_px<PxU8>(fmt, ops, n, src, dst, g, tune, stream);
    if (fmt == kFmtRGBA32F) return launch_ops_px<PxF32>(fmt, ops, n, src, dst, g, tune, stream);
    return hipErrorInvalidValue;
}

}  // namespace rf
