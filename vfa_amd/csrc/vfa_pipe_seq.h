// vfa_pipe_seq.h -- the order in which a workgroup of the pipelined frame kernel (vfa_pipe.hip) walks its share of the frame,
// as plain host / device code: tests/native/pipe_seq_harness.cpp compiles it with g++ and checks it on the CPU.
//
// Work of a frame, for ANY number of z-layers (reference vfa/model/vfa_op.py:50-59, 118-125: K = nl * C):
//   tile     = 8 x 4 BEV cells = one 32-row block of the matrix pipe
//   sub-tile = one live view of one (tile, scale): 32 rows x 256 columns of accumulators
//   run      = rt consecutive tiles, rt = 1, 2 or 4 (run_tiles_of: chosen per frame; rt = 1 is the order of rounds 3-5, tile by tile)
//   group    = up to four sub-tiles of one (run, scale), taken in (tile, view) order ACROSS the tiles of the run (round 6): their
//              accumulators (4 x 32 rows x 256 columns, fp32) stay in registers over all layers, because `relu` follows the sum over
//              the WHOLE K = nl * 256 (vfa_op.py:123-124), while `collapse.weight` streams through once per group and layer.  Until
//              round 5 a group held views of ONE tile: a rig of seven cameras ran groups of 4 + 3 (every eighth sub-tile slot empty),
//              one of six 4 + 2, and a rank that holds ONE camera of a sharded rig streamed the whole weight for 32 rows.  Now the
//              28 sub-tiles of a seven-camera run are seven full groups, and a one-camera frame fills a group with four tiles.
//   phase    = (group, layer)
//   step     = (phase, channel quarter q, set): set 0 = sub-tiles 0, 1 of the group, set 1 = sub-tiles 2, 3; one step is 64 rows x
//              64 channels of voxel features, pooled by the pooling waves while the eight matrix waves multiply the
//              previous step; both sets of a (layer, quarter) use the same 64 x 256 slice of `collapse.weight`.
//              EVERY (phase, quarter) has both steps, also when the group has no third sub-tile (the step is then empty: a barrier
//              and nothing else): the set of a step is the parity of its index, so the kernel's loop, unrolled by two, addresses
//              the accumulators of each set statically.
// Order: run -> scale -> groups -> layers.  A workgroup owns the groups from the k_begin-th group of run r_begin up to (not
// including) the k_end-th group of run r_end (groups of a run numbered across its scales).
#ifndef VFA_PIPE_SEQ_H
#define VFA_PIPE_SEQ_H

#if defined(__HIPCC__)
#define VFA_SEQ_HD __host__ __device__ __forceinline__
#else
#define VFA_SEQ_HD inline
#endif

namespace vfa_pipe {

constexpr int kSeqMaxScales = 3;
constexpr int kGroupViews = 4;             // sub-tiles of a group
constexpr int kRunTiles = 4;               // the LARGEST run (the tile offset of a sub-tile takes two bits of its byte); a frame's run length is `rt`

VFA_SEQ_HD int seq_popc(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}
VFA_SEQ_HD int seq_ctz(unsigned v) { return __builtin_ctz(v); }

// groups of a (run, scale) that holds `n_sub` live sub-tiles
VFA_SEQ_HD int groups_of_count(int n_sub) { return (n_sub + kGroupViews - 1) / kGroupViews; }
VFA_SEQ_HD int runs_of(int n_tiles, int rt) { return (n_tiles + rt - 1) / rt; }

// Run length of a frame (measured, round 6, launch time of the frame kernel in us on one MI355X, rt = 1 / 2 / 4; round 5 = tile by tile):
//   seven cameras, MultiviewC 156 x 156 x 5     1 456-1 488 / 1 470-1 490 / 1 468-1 475   (round 5: 1 445-1 575)
//   six cameras,   MultiviewX 160 x 250 x 8     3 376-3 381 / 3 125-3 136 / 3 135-3 149    (3 400-3 420)
//   seven cameras, Wildtrack 120 x 360 x 8      3 746-3 864 / 3 666-3 719 / 3 570-3 595    (3 695-3 790)
//   eight cameras, 512 x 512 x 32, one band     27 100 / 25 250 / 23 300-23 350            (26 600-26 750)
//   one camera,    512 x 512 x 32               28 500 / 19 000 / 13 400                   (four-step phase of rounds 4-5: 20 700-20 850)
//   one camera,    Wildtrack 120 x 360 x 8      606 (four-step) / 646 / 596                (625)
//   one camera,    MultiviewC 156 x 156 x 5     285 (four-step) / 314 / 317                (305)
//   two cameras,   MultiviewC 156 x 156 x 5     488 (four-step) / 474 / 494                (485)
// Longer runs fill the groups (fewer steps: -12 % for seven cameras, -25 % for six, -75 % for one) and keep a workgroup on one
// image for longer (the eight-camera frame has nothing to fill and still gains 12 %); they cost at the end of a run -- every
// workgroup that holds groups of a run reads its contributions to ALL tiles of the run back, and a group, the unit of the work cuts,
// gets no smaller --: what a frame can afford grows with the steps a workgroup has between two ends of runs.  Rigs of three and more
// cameras on small frames keep the tile-by-tile order and its register-carried sums (pipe_kernel<.., RT1>); frames of one or two
// views are a rank's share of a camera-sharded rig: big ones fill their groups from four tiles, small ones keep the four-step phase
// (rt = 1, two sub-tiles per group at most: pipe_kernel<.., SMALL>).
VFA_SEQ_HD int run_tiles_of(int n_views, int n_tiles, int n_scales, int nl, int n_blocks)
{
    const long long steps = 2ll * nl * n_views * n_scales * n_tiles / (n_blocks > 0 ? n_blocks : 1); // per workgroup, full groups
    if (n_views <= 2) return steps >= 250 ? kRunTiles : 1;
    return steps >= 1600 ? kRunTiles : (steps >= 1200 ? 2 : 1);
}

// sub-tile j of a group in byte j of `subs`: view in bits 0-4, tile offset inside the run in bits 5-6
VFA_SEQ_HD unsigned sub_byte(int view, int tile_off) { return (unsigned)view | ((unsigned)tile_off << 5); }
VFA_SEQ_HD int sub_view(unsigned subs, int j) { return (int)((subs >> (8 * j)) & 31u); }
VFA_SEQ_HD int sub_tile_off(unsigned subs, int j) { return (int)((subs >> (8 * j + 5)) & 3u); }

struct Step {
    int run;         // < 0: no step
    int scale, layer, q, set;
    int nj;          // sub-tiles of the group, 1..4
    unsigned subs;   // sub-tile j in byte j (sub_byte)
    int phase;       // running number of the (group, layer) of this step within the workgroup
    int rank;        // index of the group among the groups of its run
    int ci;          // contribution index of the group's head tile (walk_groups)
    int index;       // running number of the step within the workgroup (parity = LDS buffer)
    bool grp_first;  // first step of the group for this set: the accumulators of the set start from the bias
    bool grp_last;   // last step of the group for this set: relu and the view sum follow
    bool run_last;   // last step of this workgroup's part of the run
    VFA_SEQ_HD bool valid() const { return run >= 0; }
    VFA_SEQ_HD int view(int j) const { return sub_view(subs, j); }
    VFA_SEQ_HD int tile(int j, int rt) const { return run * rt + sub_tile_off(subs, j); }
    VFA_SEQ_HD int in_set() const { return nj - 2 * set >= 2 ? 2 : (nj - 2 * set > 0 ? nj - 2 * set : 0); } // sub-tiles of this step: 0, 1 or 2
};

// A phase = (group, layer): eight steps, quarter q = k >> 1, set = k & 1 for k = 0 .. 7.  The kernel hands PHASES from the one
// wave that runs the generator to the others (through LDS) and every wave derives the steps itself (step_of).
struct Phase {
    int run;         // < 0: no phase
    int scale, layer, nj;
    unsigned subs;
    int phase, rank, ci;
    bool more_in_run; // another group of this workgroup follows in the same run
    VFA_SEQ_HD bool valid() const { return run >= 0; }
};

VFA_SEQ_HD Step step_of(const Phase &ph, int k, int nl, int index)
{
    Step st;
    st.run = ph.run; st.scale = ph.scale; st.layer = ph.layer; st.q = k >> 1; st.set = k & 1; st.nj = ph.nj; st.subs = ph.subs;
    st.phase = ph.phase; st.rank = ph.rank; st.ci = ph.ci; st.index = index;
    st.grp_first = ph.layer == 0 && st.q == 0;
    st.grp_last = ph.layer == nl - 1 && st.q == 3;
    st.run_last = st.grp_last && st.set == 1 && !ph.more_in_run;
    return st;
}

// The sums of a tile.  A group ends with relu and the view sum (vfa_op.py:124; vfanet.py:79, 82): the sub-tiles of one tile inside a
// group are added in registers and the result -- one CONTRIBUTION per (group, tile) -- is stored to a buffer of the workgroup's own,
// [tile of the run][scale][index]; when the workgroup leaves the run it adds the contributions of every tile in (scale, index) order.
// A tile that enters a group behind another tile starts at index 0; the HEAD tile of a group (its first sub-tile's) continues from
// the group before when that one ended in the same tile: index + 1.  A tile of n live views touches at most (n + 2) / 4 + 1 groups.
VFA_SEQ_HD int contributions_of(int n_views) { return (n_views + 2) / 4 + 1; }

// Walks the groups of ONE run: emit(k, scale, subs, nj, ci) for the k-th group, in the kernel's order; ci = the contribution index
// of the group's head tile (every other tile of the group has index 0).  `mask(s, off)` = live-view mask of (scale s, tile run *
// rt + off), 0 beyond the last tile; `lo` = the first group the caller's workgroup owns in this run (its indices start there).
// Returns the number of groups.  The generator of the kernel (vfa_pipe.hip: fill_groups) and the CPU harness enumerate through
// this one function; the work cuts count the same groups from the live sub-tiles per scale (walk_run below).
template <class Mask, class Emit>
VFA_SEQ_HD int walk_groups(int n_scales, int rt, int lo, Mask &&mask, Emit &&emit)
{
    int k = 0;
    for (int s = 0; s < n_scales; ++s) {
        unsigned subs = 0;
        int nj = 0, last_off = -1, last_ci = 0; // the tile the group before ended in (this scale), and its contribution index there
        auto flush = [&]() {
            const int head = sub_tile_off(subs, 0), tail = sub_tile_off(subs, nj - 1);
            const int ci = (k != lo && head == last_off) ? last_ci + 1 : 0;
            emit(k, s, subs, nj, ci);
            last_off = tail;
            last_ci = tail == head ? ci : 0;
            ++k; subs = 0; nj = 0;
        };
        for (int off = 0; off < rt; ++off) {
            unsigned rest = mask(s, off);
            while (rest) {
                subs |= sub_byte(seq_ctz(rest), off) << (8 * nj);
                rest &= rest - 1u;
                if (++nj == kGroupViews) flush();
            }
        }
        if (nj) flush();
    }
    return k;
}

// Generator of the phases / steps of one workgroup (CPU harness; the kernel expands runs into a ring of group records with
// walk_groups and counts the layers itself).  `Masks` returns the live-view mask of (scale, tile).
template <class Masks>
struct Sequencer {
    Masks masks;
    int n_scales, nl, n_tiles, rt, r_begin, k_begin, r_end, k_end, r_lim;
    int run, n_groups, next_k, k_hi, layer, q, phase, index;
    // the groups of the current run (at most 3 scales x 32 views x 4 tiles / 4)
    int g_scale[96], g_nj[96], g_ci[96];
    unsigned g_subs[96];
    bool in_group;

    void load_run()
    {
        n_groups = 0;
        if (run >= r_lim) return;
        const int base = run * rt;
        next_k = run == r_begin ? k_begin : 0;
        n_groups = walk_groups(n_scales, rt, next_k, [&](int s, int off) { return base + off < n_tiles ? masks(s, base + off) : 0u; },
                               [&](int k, int s, unsigned subs, int nj, int ci) { g_scale[k] = s; g_subs[k] = subs; g_nj[k] = nj; g_ci[k] = ci; });
        k_hi = run == r_end ? (k_end < n_groups ? k_end : n_groups) : n_groups;
    }
    void begin(int n_scales_, int nl_, int n_tiles_, int rt_, int r_begin_, int k_begin_, int r_end_, int k_end_)
    {
        n_scales = n_scales_; nl = nl_; n_tiles = n_tiles_; rt = rt_; r_begin = r_begin_; k_begin = k_begin_; r_end = r_end_; k_end = k_end_;
        r_lim = k_end > 0 ? r_end + 1 : r_end;
        run = r_begin; layer = 0; q = 0; phase = -1; index = -1; in_group = false; cur.run = -1;
        load_run();
    }
    Phase next_phase()
    {
        Phase ph;
        ph.run = -1; ph.scale = ph.layer = ph.nj = 0; ph.subs = 0; ph.phase = ph.rank = ph.ci = 0; ph.more_in_run = false;
        if (in_group) {
            if (layer + 1 < nl) ++layer;
            else { in_group = false; ++next_k; }
        }
        if (!in_group) {
            while (run < r_lim && next_k >= k_hi) { ++run; load_run(); }
            if (run >= r_lim) return ph;
            in_group = true;
            layer = 0;
        }
        ++phase;
        ph.run = run; ph.scale = g_scale[next_k]; ph.layer = layer; ph.nj = g_nj[next_k]; ph.subs = g_subs[next_k]; ph.phase = phase;
        ph.rank = next_k; ph.ci = g_ci[next_k]; ph.more_in_run = next_k + 1 < k_hi;
        return ph;
    }
    Phase cur;
    Step next()
    {
        if (index < 0 || q == 7) { cur = next_phase(); q = 0; }
        else ++q; // (q doubles as the step counter inside the phase here)
        if (!cur.valid()) { Step st; st.run = -1; st.scale = st.layer = st.q = st.set = st.nj = 0; st.subs = 0; st.phase = st.rank = st.ci = st.index = 0;
                            st.grp_first = st.grp_last = st.run_last = false; q = 7; index = index < 0 ? 0 : index; return st; }
        ++index;
        return step_of(cur, q, nl, index);
    }
};

// ------------------------------------------------------------------------------------------------
// work cuts: the groups of the frame in (run, scale, tile, view) order, cut into `n_chunks` pieces of equal estimated cost.
// Cost model, round 6 (tools/fit_pipe_cost.py: least squares of the cycles every workgroup of the production kernel took against what
// it had to do, on the three shipped configs; the same terms within 1 % on all of them), in units of 16 cycles:
//   per (group, layer), its eight steps whatever is in them (barriers, tables, weight slices)   625  (+ 100 when set 1 is empty)
//   per sub-tile and layer with a live box (four quarter-steps of pooling + products)           366  + 2 per slot of its tap window
//                                                                                                 or + 500 when it is pooled from L2
//   per sub-tile and layer without one (the planes are zeroed, the products run)                188
//   per group 9, per tile of the run (its store) 144 on the run's first group
// The geometry pass (pipe_records_kernel) adds the per-layer terms of a sub-tile up: `subcost[scale][tile * n_views + view]`.
// Rounds 3-5 priced a group by its view count alone (+ the tile's L2 items on its first group): heaviest workgroup 1.05-1.08 x the
// mean on the shipped configs; this model leaves 1.01-1.02 in the fit.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kPhaseCost = 625, kPhaseSmallExtra = 100, kSubLiveCost = 366, kSubSlotCost = 2, kSubDirectCost = 500, kSubDeadCost = 188;
constexpr unsigned kGroupCost = 9, kTileCost = 144, kEmptyCost = 16;

VFA_SEQ_HD unsigned sub_layer_cost(bool live, bool direct, int n_slots)
{
    return live ? kSubLiveCost + (direct ? kSubDirectCost : kSubSlotCost * (unsigned)n_slots) : kSubDeadCost;
}

// walks the groups of a run: visit(k, w0, w1) for the k-th group covering [w0, w1) of the run's cost; `mask(s, off)` as in
// walk_groups, `subcost(s, off, view)` = the sub-tile's cost over all layers; `tiles` = tiles of the run.  Returns the run's cost.
template <class Mask, class SubCost, class Visit>
VFA_SEQ_HD unsigned walk_run(int n_scales, int rt, int nl, int tiles, Mask &&mask, SubCost &&subcost, Visit &&visit)
{
    unsigned w = 0;
    const int k = walk_groups(n_scales, rt, 0, mask, [&](int kk, int s, unsigned subs, int nj, int) {
        unsigned wi = (unsigned)nl * (kPhaseCost + (nj <= 2 ? kPhaseSmallExtra : 0u)) + kGroupCost;
        for (int j = 0; j < nj; ++j) wi += subcost(s, sub_tile_off(subs, j), sub_view(subs, j));
        if (kk == 0) wi += kTileCost * (unsigned)tiles;
        visit(kk, w, w + wi);
        w += wi;
    });
    return k == 0 ? kEmptyCost * (unsigned)tiles : w;
}

// The same by SCALE (the cuts kernel works on (run, scale) entries: a thread per entry instead of per run -- a run of four tiles and
// seven views is ~85 sub-tiles of strictly serial code).  emit(j, subs, nj) for the j-th group of scale s; returns their number.
template <class Mask, class Emit>
VFA_SEQ_HD int walk_scale_groups(int rt, int s, Mask &&mask, Emit &&emit)
{
    int k = 0, nj = 0;
    unsigned subs = 0;
    for (int off = 0; off < rt; ++off) {
        unsigned rest = mask(s, off);
        while (rest) {
            subs |= sub_byte(seq_ctz(rest), off) << (8 * nj);
            rest &= rest - 1u;
            if (++nj == kGroupViews) { emit(k, subs, nj); ++k; subs = 0; nj = 0; }
        }
    }
    if (nj) { emit(k, subs, nj); ++k; }
    return k;
}
// cost of the groups of scale s of a run, as walk_run counts them: `first_of_run` = no scale in front of s has a group (the run's
// first group carries the tiles' cost); visit(j, w0, w1) = the j-th group of the scale covers [w0, w1) of the scale's cost
template <class Mask, class SubCost, class Visit>
VFA_SEQ_HD unsigned walk_scale(int rt, int nl, int tiles, int s, bool first_of_run, Mask &&mask, SubCost &&subcost, Visit &&visit)
{
    unsigned w = 0;
    walk_scale_groups(rt, s, mask, [&](int j, unsigned subs, int nj) {
        unsigned wi = (unsigned)nl * (kPhaseCost + (nj <= 2 ? kPhaseSmallExtra : 0u)) + kGroupCost;
        for (int i = 0; i < nj; ++i) wi += subcost(s, sub_tile_off(subs, i), sub_view(subs, i));
        if (j == 0 && first_of_run) wi += kTileCost * (unsigned)tiles;
        visit(j, w, w + wi);
        w += wi;
    });
    return w;
}

// groups of a run from its masks alone
template <class Mask>
VFA_SEQ_HD int groups_of_run(int n_scales, int rt, Mask &&mask)
{
    int k = 0;
    for (int s = 0; s < n_scales; ++s) {
        int cnt = 0;
        for (int off = 0; off < rt; ++off) cnt += seq_popc(mask(s, off));
        k += groups_of_count(cnt);
    }
    return k;
}

} // namespace vfa_pipe
#endif // VFA_PIPE_SEQ_H
