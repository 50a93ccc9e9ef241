// Launch planner of the MFMA path: pure host C++ (no HIP header, no device call).
//
// Everything the accumulate_mfma kernel dereferences is decided here -- the slot order of a sightline's kept pixels,
// the passes, the work items, the piece counts the finalisation adds up -- so this file is compiled twice: into
// libdlagp_hip.so (dlagp_hip.hip), and alone into libdlagp_plan_host.so with `g++ -fsanitize=address,undefined`
// (plan_host_debug.cpp, `make sanitize`), where tests/test_planner_fuzz.py drives it through dlagp_plan_debug with
// random masks, sample counts and forced cuts on the CPU.
//
//   slots   the kept pixels in order, padded so that every chunk of 8 slots has its convolution windows inside
//           64 consecutive entries (a mask gap of more than ~50 pixels closes the chunk with zero-weight slots;
//           reference masks: null_gp.py:139-152, read_spec.py:38-71)
//   passes  runs of <= 7 chunks whose windows fit in 64 consecutive entries (kernels_mfma.h)
//   items   one wave each: 16 samples x a range of chunks of one sightline
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/dlagp.h"

#ifndef DLAGP_STEP_PIXELS
#define DLAGP_STEP_PIXELS 4 /* K extent of one v_mfma_f64_16x16x4_f64 (dlagp_device.h) */
#endif
#define DLAGP_MFMA_GROUP 16  // samples per wave / workgroup
#define DLAGP_PASS_ENTRIES 64
#ifndef DLAGP_CHUNK_PIXELS
// planning granule ("chunk") = two steps.  One step (4) was tried: the same number of passes on the bench workload,
// 1 % slower (profiles/r02_work_decomposition.txt)
#define DLAGP_CHUNK_PIXELS 8
#endif
#define DLAGP_MAX_PASS_STEPS 14  // 56 pixels
#define DLAGP_CHUNK_STEPS (DLAGP_CHUNK_PIXELS / DLAGP_STEP_PIXELS)
#define DLAGP_MAX_PASS_CHUNKS (DLAGP_MAX_PASS_STEPS / DLAGP_CHUNK_STEPS)
#define DLAGP_CACHED_PASS_CHUNKS 8  // cached levels of a sweep: passes of 64 slots (kernels_mfma.h) ...
#define DLAGP_CACHE_TILE_SLOTS (DLAGP_CACHED_PASS_CHUNKS * DLAGP_CHUNK_PIXELS)  // ... = the tiles of the profile cache
#define DLAGP_MAX_PIECES 255     // pieces per block of 16 samples: the finalize kernels count them in a byte

namespace dlagp {

struct PassDesc {
    int entry_start;  // first entry computed by phase 1
    int chunk_begin;  // first chunk (DLAGP_CHUNK_PIXELS slots) consumed by phase 2
    int nchunks;      // 1..DLAGP_MAX_PASS_CHUNKS
    int pad;
};

// One wave's work: 16 samples (`group`) of sightline `b` over the passes [pass_begin, pass_end) of the launch
// plan's pass list; its sums go to partial slot `part` of the block.
// `colgroup`: which slice of the sightline's columns this wave contracts (ranks above 20 run as several column groups of
// 210 vech columns + 20 columns of M each, see kernels_mfma.h; always 0 for rank <= 20).
struct WorkItem {
    int b, group, part, pass_begin, pass_end, colgroup, pad1, pad2;
};

namespace plan {

// planner tunables (library options of the same names)
struct Options {
    int64_t pixel_splits = 0;  // > 0: force that many equal pieces per block
    int64_t item_cost = 20;    // fixed cost of a work item in the model, tenths of a chunk
    int64_t rest_first = 0;    // dispatch the remainder pieces before the full ones
    int64_t slots = 2048;      // wave slots to fill: 256 CUs x 4 SIMDs x 2 waves
    int64_t xcd_affine = 1;    // batches of LONG sightlines: deal the items so that an XCD sweeps one sightline at a time
    int64_t affine_min_chunks = 256;  // ... "long": the mean sightline has at least this many chunks (2048 slots: a 4 MB operand)
    int64_t column_groups = 1; // ranks above 20: every (block, piece) is evaluated once per column group
    int64_t taper = 1;         // batches: pieces per block for the sightlines of the last round of eight (1 = off)
};

// what the planner needs to know of one sightline
struct Sightline {
    const std::vector<int> *slot_pos;  // window start of every slot (plan_slots)
    int ntaps;
    long chunks;                       // slots / DLAGP_CHUNK_PIXELS
};

struct HostPlan {
    std::vector<PassDesc> passes;        // pass lists of all pieces (shared between the groups of a sightline)
    std::vector<WorkItem> items;         // dispatch order
    std::vector<unsigned char> counts;   // [nb * ngroups] pieces per block
    int max_parts = 1;
    int full = 1;                        // the cut that was chosen: `full` pieces of `frac` of a sightline's chunks
    double frac = 1.0;
    bool remainder = false;              // ... plus a remainder piece
};

// window start of kept pixel i = index of its unmasked pixel: with broadening the padded grid has 3 extra entries in
// front, so the 7-tap "valid" window of unmasked pixel j is entries j .. j + 6 (voigt.py:319, dla_gp.py:388)
inline void window_starts(const uint8_t *keep, int n_u, std::vector<int> &pos) {
    pos.clear();
    for (int j = 0; j < n_u; ++j)
        if (keep[j]) pos.push_back(j);
}

inline void plan_slots(const std::vector<int> &pos, int n, int ntaps, std::vector<int> &slot_pixel,
                       std::vector<int> &slot_pos) {
    slot_pixel.clear();
    slot_pos.clear();
    int i = 0;
    while (i < n) {
        const int first = pos[i];
        int filled = 0;
        while (i < n && filled < DLAGP_CHUNK_PIXELS && pos[i] + ntaps <= first + DLAGP_PASS_ENTRIES) {
            slot_pixel.push_back(i);
            slot_pos.push_back(pos[i]);
            ++i;
            ++filled;
        }
        for (; filled < DLAGP_CHUNK_PIXELS; ++filled) {  // zero-weight slots: they re-read the last pixel's window
            slot_pixel.push_back(-1);
            slot_pos.push_back(slot_pos.back());
        }
    }
}

// passes of the chunk range [c0, c1) of one sightline, appended to `passes`
inline void plan_passes(const std::vector<int> &slot_pos, int ntaps, int c0, int c1, std::vector<PassDesc> &passes) {
    int c = c0;
    while (c < c1) {
        PassDesc pd{slot_pos[(size_t)c * DLAGP_CHUNK_PIXELS], c, 1, 0};  // a single chunk always fits (plan_slots)
        while (c + pd.nchunks < c1 && pd.nchunks < DLAGP_MAX_PASS_CHUNKS &&
               slot_pos[(size_t)(c + pd.nchunks) * DLAGP_CHUNK_PIXELS + DLAGP_CHUNK_PIXELS - 1] + ntaps <=
                   pd.entry_start + DLAGP_PASS_ENTRIES)
            ++pd.nchunks;
        passes.push_back(pd);
        c += pd.nchunks;
    }
}

// Work items of one launch.  A wave owns 16 samples ("group") of one sightline over a range of its chunks; the
// chip holds 2048 such waves at a time (256 CUs x 4 SIMDs x 2, register-limited), and a SIMD's FP64 pipe is
// busy for the sum of its waves' work.  Every group of a sightline is cut at the SAME chunk positions: the waves
// then sweep the sightline's operand Pm (6 MB, more than one XCD's L2) as a few coherent fronts that stay
// L2-resident -- cutting the (group, chunk) space into 2048 equal intervals balances perfectly but puts every wave
// at a different position and came out 10 % slower (profiles/r02_work_decomposition.txt).  Two shapes of cut are
// priced with a simple model (time of the busiest wave slot, in chunks; `o` = fixed cost of an item) and the
// cheaper one is taken:
//   (A) h equal pieces per group, h = 1..8:            ceil(groups h / 2048) (C / h + o)   [+ finalisation, ~ groups h]
//   (B) f = floor(2048 / groups) full pieces of F chunks plus a remainder piece R = C - f F; the 2048 - f groups
//       slots left over work through the remainder pieces, m = ceil(groups / slots left) each:
//                                                        F + o = m (R + o)
//       One sightline x 10 000 samples = 625 groups: (A) h = 3 leaves 173 of the 2048 slots empty (129 chunks per
//       slot); (B) f = 3, F = 117, R = 24 fills them (122).
// Items are dispatched in index order as slots free up: full pieces first, then the remainders; the order affects
// only the balance, never the results.  Every block of 16 samples adds its pieces' partial sums in piece order
// (finalize kernels, `counts`), so results are deterministic.  Option pixel_splits = h forces (A) with that h.
// Modelled time of one work item, in units of one chunk's weights + MFMA work (measured with the per-wave trace of
// tools/wave_timeline.py: 7.2 us per item + 6.6 us per pass + 1.3 us per chunk): the Voigt profiles, the convolution
// and the set-up of a pass cost the same whether the pass holds 7 chunks or 1.
inline double item_cost(const std::vector<int> &slot_pos, int ntaps, int c0, int c1, const Options &opt) {
    if (c1 <= c0) return 0.0;
    std::vector<PassDesc> passes;
    plan_passes(slot_pos, ntaps, c0, c1, passes);
    constexpr double per8 = 8.0 / DLAGP_CHUNK_PIXELS;  // the constants below were measured per 8 slots
    return per8 * (0.1 * (double)opt.item_cost + 5.1 * (double)passes.size()) + (double)(c1 - c0);
}

// Moves the cuts of one sightline (edge[0] = 0 < edge[1] < ... < edge[last] = chunks) by up to 24 chunks each so that
// the busiest wave slot gets as little as possible: in practice onto pass boundaries, so that no piece pays the
// fixed cost of a pass for a handful of chunks.  The last piece is the remainder when `remainder`; a left-over
// slot works through `rest_per_slot` of those.
inline void refine_cuts(const std::vector<int> &slot_pos, int ntaps, std::vector<int> &edge, bool remainder,
                        long rest_per_slot, const Options &opt) {
    const int npieces = (int)edge.size() - 1;
    if (npieces < 2) return;
    auto piece_cost = [&](int j) {
        const double c = item_cost(slot_pos, ntaps, edge[j], edge[j + 1], opt);
        return remainder && j == npieces - 1 ? c * (double)rest_per_slot : c;
    };
    std::vector<double> cost(npieces);
    for (int j = 0; j < npieces; ++j) cost[j] = piece_cost(j);
    for (int round = 0; round < 3; ++round) {
        bool moved = false;
        for (int j = 1; j < npieces; ++j) {  // cut between pieces j - 1 and j
            const int at = edge[j];
            int best_at = at;
            double best = std::max(cost[j - 1], cost[j]), best_sum = cost[j - 1] + cost[j];
            for (int d = -24; d <= 24; ++d) {
                const int cand = at + d;
                if (d == 0 || cand <= edge[j - 1] || cand >= edge[j + 1]) continue;
                edge[j] = cand;
                const double a = piece_cost(j - 1), bb = piece_cost(j);
                const double worst = std::max(a, bb);
                if (worst < best - 1e-9 || (worst < best + 1e-9 && a + bb < best_sum - 1e-9)) {
                    best = worst;
                    best_sum = a + bb;
                    best_at = cand;
                }
            }
            edge[j] = best_at;
            if (best_at != at) {
                moved = true;
                cost[j - 1] = piece_cost(j - 1);
                cost[j] = piece_cost(j);
            }
        }
        if (!moved) break;
    }
}

// the cut, as fractions of a sightline's chunks: `full` pieces of `frac` each, then a remainder if frac * full < 1
inline void choose_cut(long groups, double C, const Options &opt, int *full_out, double *frac_out) {
    const long slots = std::max<int64_t>(opt.slots, 1);
    const int64_t forced = opt.pixel_splits;
    // prologue + epilogue of an item, and the pass it adds by starting a new one, in chunks' worth of work
    const double o = 0.1 * (double)opt.item_cost * 8.0 / DLAGP_CHUNK_PIXELS;
    // every piece of a block is one more set of partial sums the finalisation reads: ~25 ns per (block, piece) in a
    // large launch, in units of a wave slot's time per chunk (4.9 us per 8 slots)
    const double finalize_per_piece = 0.005 * 8.0 / DLAGP_CHUNK_PIXELS;
    int full = 1;
    double frac = 1.0, best = 1e300;
    // up to 32 equal pieces (8 until round 4): a launch of few blocks -- one rank's S / 8 = 1250 samples when ONE
    // sightline's samples are split over 8 GPUs: 79 blocks -- filled a third of the wave slots with 8 pieces each
    const int h_max = (int)std::max(1.0, std::min(32.0, std::floor(C / DLAGP_MAX_PASS_CHUNKS)));
    for (int h = 1; h <= h_max; ++h) {
        if (forced > 0 && h != std::min<int64_t>(forced, h_max)) continue;
        const double cost = (double)((groups * h + slots - 1) / slots) * (C / h + o) + finalize_per_piece * groups * h;
        if (cost < best) {
            best = cost;
            full = h;
            frac = 1.0 / h;
        }
    }
    // at most 254 full pieces + the remainder: the finalize kernels count a block's pieces in a byte (`counts`)
    const long f = std::min<long>(slots / std::max<long>(groups, 1), DLAGP_MAX_PIECES - 1);
    if (forced <= 0 && f >= 1 && slots - f * groups > 0) {
        const long m = (groups + (slots - f * groups) - 1) / (slots - f * groups);
        const double F = (m * (C + o) - o) / (1.0 + (double)m * f);
        const double R = C - f * F;
        if (R >= DLAGP_MAX_PASS_CHUNKS && F + o + finalize_per_piece * groups * (f + 1) < best) {
            full = (int)f;
            frac = F / C;
        }
    }
    *full_out = full;
    *frac_out = frac;
}

// The launch plan of `S` samples of every sightline.  Returns 0, or DLAGP_EINVAL with `err` set.
inline int build(const std::vector<Sightline> &sightlines, int64_t S, const Options &opt, HostPlan &plan,
                 std::string &err) {
    const int nb = (int)sightlines.size();
    if (nb < 1 || S < 1) {
        err = "planner: need at least one sightline and one sample";
        return DLAGP_EINVAL;
    }
    const long ngroups = (long)((S + DLAGP_MFMA_GROUP - 1) / DLAGP_MFMA_GROUP);
    const int colgroups = (int)std::max<int64_t>(1, opt.column_groups);
    // a block of 16 samples with G column groups fills wave slots like G blocks: the cut is chosen for that many
    const long groups = (long)nb * ngroups * colgroups;
    const long blocks = (long)nb * ngroups;
    double mean_chunks = 0.0;
    for (int b = 0; b < nb; ++b) {
        if (sightlines[b].chunks < 1 ||
            (size_t)sightlines[b].chunks * DLAGP_CHUNK_PIXELS != sightlines[b].slot_pos->size()) {
            err = "planner: a sightline's slot table is not a whole number of chunks";
            return DLAGP_EINVAL;
        }
        mean_chunks += (double)sightlines[b].chunks / nb;
    }
    int full = 1;
    double frac = 1.0;
    choose_cut(groups, mean_chunks, opt, &full, &frac);
    struct Piece {
        int b, group, part, pass_begin, pass_end;
        bool rest;
    };
    std::vector<PassDesc> &passes = plan.passes;
    passes.clear();
    const bool remainder = frac * full < 1.0 - 1e-9;
    plan.counts.assign((size_t)blocks, 1);
    int max_parts = 1;
    // remainder items a left-over wave slot works through (shape (B)); 1 otherwise
    const long slots_total = std::max<int64_t>(opt.slots, 1), left = slots_total - (long)full * groups;
    const long rest_per_slot = remainder && left > 0 ? (groups + left - 1) / left : 1;
    // XCD affinity (batches, one piece per block): workgroups are dealt round-robin over the 8 XCDs (observed, for speed
    // only), each with its own 4 MB L2, and a sightline's MFMA operand is 6 MB.  In (sightline, group) order every XCD
    // has waves of the ~3 sightlines that are in flight at a time; dealt so that workgroup i works on a sightline
    // b = i (mod 8), every XCD sweeps ONE sightline's operand with all of its waves.  Option "plan_xcd_affine".
    // Round 5: only where the operands are LONG -- one alone at least an XCD's 4 MB of L2 (n_slots >= 2048; the bench's
    // n = 3000: 6 MB).  On DR12Q-sized sightlines (operand <= 2.5 MB) the plain (sightline, block) order -- every XCD sees
    // the same ~3 sightlines in flight -- is within 1 % of the affine order at equal lengths (n = 685 / 1250) and 12 %
    // FASTER on a ragged batch (32 sightlines, n_u uniform in 264 ... 1246: 0.50 -> 0.58 of peak;
    // profiles/r05_experiments.txt item 8): with the sightlines sorted by length, XCD 0 got the longest member of EVERY
    // round of eight and XCD 7 the shortest -- 15 % more work on one XCD than on average, and a launch that ends when that
    // XCD does.  (A per-item trace, tools/trace_ragged.py, shows every XCD's wave slots 99.8 % occupied while items keep
    // coming: nobody waits for anybody in the middle of the launch -- the loss is the tail.)  The affine rounds therefore
    // also alternate their direction (below), so that a long operand's batch stays balanced when its lengths differ.
    const bool batch_of_blocks = !remainder && full == 1 && nb >= 8;
    const bool affine = opt.xcd_affine != 0 && batch_of_blocks && mean_chunks >= (double)opt.affine_min_chunks;
    // Longest sightlines first (a catalogue batch is ragged: n follows z_qso): what is in flight at any moment is of similar
    // length -- the eight of an affine round, the ~3 sightlines of the plain order -- and the launch ends on its shortest
    // items instead of on whatever came last.  Equal lengths (the bench) keep their order.
    std::vector<int> order(nb);
    for (int b = 0; b < nb; ++b) order[b] = b;
    if (batch_of_blocks) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sightlines[x].chunks > sightlines[y].chunks; });
    // Taper: a launch of many equal rounds ends with every wave slot somewhere inside its last item -- on average half an
    // item of idle time per slot.  The sightlines of the LAST round of eight are cut into `taper` pieces each, so that
    // the launch ends on items a fraction as long.  Option "plan_taper" (0 / 1: off).
    const int taper = affine && nb >= 16 && opt.taper > 1 ? (int)std::min<int64_t>(opt.taper, 8) : 1;
    const int taper_from = taper > 1 ? (nb - 1) / 8 * 8 : nb;  // position in `order` of the last round's first sightline
    std::vector<int> position(nb);
    for (int x = 0; x < nb; ++x) position[order[x]] = x;
    std::vector<std::vector<Piece>> cuts_of(nb);
    for (int b = 0; b < nb; ++b) {
        // cut positions of sightline b (the same for all of its groups); empty pieces are dropped
        const long chunks = sightlines[b].chunks;
        const bool tapered = position[b] >= taper_from && chunks >= (long)taper * DLAGP_MAX_PASS_CHUNKS;
        const int npieces = tapered ? taper : full + (remainder ? 1 : 0);
        const double piece_frac = tapered ? 1.0 / taper : frac;
        std::vector<int> edge(npieces + 1, (int)chunks);
        edge[0] = 0;
        for (int j = 1; j < npieces; ++j)
            edge[j] = (int)std::max<long>(edge[j - 1], std::min<long>(chunks, std::llround(j * piece_frac * chunks)));
        refine_cuts(*sightlines[b].slot_pos, sightlines[b].ntaps, edge, remainder && !tapered, rest_per_slot, opt);
        std::vector<Piece> &cuts = cuts_of[b];
        for (int j = 0; j < npieces; ++j)
            if (edge[j + 1] > edge[j]) {
                const int begin = (int)passes.size();  // the pass list of a piece is shared by all groups
                plan_passes(*sightlines[b].slot_pos, sightlines[b].ntaps, edge[j], edge[j + 1], passes);
                cuts.push_back({b, 0, (int)cuts.size(), begin, (int)passes.size(), remainder && !tapered && j == npieces - 1});
            }
        if (cuts.size() > DLAGP_MAX_PIECES) {
            err = "internal: " + std::to_string(cuts.size()) + " pieces per block exceed the byte-sized piece count";
            return DLAGP_EINVAL;
        }
        max_parts = std::max(max_parts, (int)cuts.size());
        for (long grp = 0; grp < ngroups; ++grp) plan.counts[(size_t)b * ngroups + grp] = (unsigned char)cuts.size();
    }
    std::vector<Piece> pieces;
    pieces.reserve((size_t)blocks * (full + 1));
    auto emit = [&](int bb, long grp) {
        for (Piece pc : cuts_of[bb]) {
            pc.group = (int)grp;
            pieces.push_back(pc);
        }
    };
    if (affine) {
        // rounds of 8 sightlines, inside a round group-major; odd rounds in reverse (the sightlines are sorted by length:
        // the XCD that gets a round's longest member gets the next round's shortest)
        for (int b0 = 0; b0 < nb; b0 += 8) {
            const int width = std::min(8, nb - b0);
            const bool reversed = (b0 / 8) % 2 == 1;
            for (long grp = 0; grp < ngroups; ++grp)
                for (int x = 0; x < width; ++x) emit(order[b0 + (reversed ? width - 1 - x : x)], grp);
        }
    } else {
        // the full pieces in (sightline, group, piece) order -- a batch's sightlines longest first -- then the remainders
        for (int x = 0; x < nb; ++x)
            for (long grp = 0; grp < ngroups; ++grp) emit(order[x], grp);
        const bool rest_first = opt.rest_first != 0;
        if (remainder) std::stable_partition(pieces.begin(), pieces.end(), [=](const Piece &x) { return x.rest == rest_first; });
    }
    // column groups one after the other: every phase of the launch sweeps ONE slice of the sightlines' operands
    plan.items.clear();
    plan.items.reserve(pieces.size() * colgroups);
    for (int cg = 0; cg < colgroups; ++cg)
        for (const Piece &pc : pieces) plan.items.push_back(WorkItem{pc.b, pc.group, pc.part, pc.pass_begin, pc.pass_end, cg, 0, 0});
    plan.max_parts = max_parts;
    plan.full = full;
    plan.frac = frac;
    plan.remainder = remainder;
    return DLAGP_OK;
}

// Body of the C entry point dlagp_plan_debug (include/dlagp.h): slot tables and launch plan of a batch of masks.
inline int debug(int nb, const int *n_u, const uint8_t *const *keep, int ntaps, int64_t S, const int64_t *options,
                 int64_t *summary, int32_t *slot_pixel_out, int32_t *slot_pos_out, int64_t slots_cap, int32_t *passes_out,
                 int64_t passes_cap, int32_t *items_out, int64_t items_cap, uint8_t *counts_out, int64_t counts_cap,
                 std::string &err) {
    if (nb < 1 || !n_u || !keep || !summary || S < 1 || (ntaps != 1 && ntaps != 7)) {
        err = "dlagp_plan_debug: need nb >= 1, S >= 1, ntaps 1 or 7, and the mask / summary arrays";
        return DLAGP_EINVAL;
    }
    Options opt;
    if (options) {
        opt.pixel_splits = options[0];
        opt.item_cost = options[1];
        opt.rest_first = options[2];
        opt.slots = options[3];
        opt.xcd_affine = options[4];
        opt.column_groups = options[5];
        opt.taper = options[6];
        if (opt.slots < 1 || opt.item_cost < 0 || opt.pixel_splits < 0 || opt.column_groups < 1 || opt.column_groups > 16 || opt.taper < 0 || opt.taper > 8) {
            err = "dlagp_plan_debug: options out of range";
            return DLAGP_EINVAL;
        }
    }
    std::vector<std::vector<int>> slot_pixel(nb), slot_pos(nb);
    std::vector<Sightline> sightlines(nb);
    int64_t total_slots = 0;
    for (int b = 0; b < nb; ++b) {
        if (n_u[b] < 1 || !keep[b]) {
            err = "dlagp_plan_debug: empty mask";
            return DLAGP_EINVAL;
        }
        std::vector<int> pos;
        window_starts(keep[b], n_u[b], pos);
        if (pos.empty()) {
            err = "dlagp_plan_debug: a mask keeps no pixel";
            return DLAGP_EINVAL;
        }
        plan_slots(pos, (int)pos.size(), ntaps, slot_pixel[b], slot_pos[b]);
        sightlines[b] = Sightline{&slot_pos[b], ntaps, (long)(slot_pos[b].size() / DLAGP_CHUNK_PIXELS)};
        total_slots += (int64_t)slot_pos[b].size();
    }
    HostPlan plan;
    const int rc = build(sightlines, S, opt, plan, err);
    if (rc) return rc;
    summary[0] = total_slots;
    summary[1] = (int64_t)plan.passes.size();
    summary[2] = (int64_t)plan.items.size();
    summary[3] = (int64_t)plan.counts.size();
    summary[4] = plan.max_parts;
    summary[5] = plan.full;
    summary[6] = plan.remainder ? 1 : 0;
    summary[7] = (int64_t)std::llround(plan.frac * 1e9);
    if (slot_pixel_out && slot_pos_out && slots_cap >= total_slots) {
        int64_t at = 0;
        for (int b = 0; b < nb; ++b)
            for (size_t i = 0; i < slot_pos[b].size(); ++i, ++at) {
                slot_pixel_out[at] = slot_pixel[b][i];
                slot_pos_out[at] = slot_pos[b][i];
            }
    }
    if (passes_out && passes_cap >= (int64_t)plan.passes.size())
        for (size_t i = 0; i < plan.passes.size(); ++i) {
            passes_out[3 * i + 0] = plan.passes[i].entry_start;
            passes_out[3 * i + 1] = plan.passes[i].chunk_begin;
            passes_out[3 * i + 2] = plan.passes[i].nchunks;
        }
    if (items_out && items_cap >= (int64_t)plan.items.size())
        for (size_t i = 0; i < plan.items.size(); ++i) {
            const WorkItem &w = plan.items[i];
            items_out[6 * i + 0] = w.b;
            items_out[6 * i + 1] = w.group;
            items_out[6 * i + 2] = w.part;
            items_out[6 * i + 3] = w.pass_begin;
            items_out[6 * i + 4] = w.pass_end;
            items_out[6 * i + 5] = w.colgroup;
        }
    if (counts_out && counts_cap >= (int64_t)plan.counts.size())
        std::copy(plan.counts.begin(), plan.counts.end(), counts_out);
    return DLAGP_OK;
}

}  // namespace plan
}  // namespace dlagp
