// vfa_pipe_seq.h -- the order in which a workgroup of the pipelined frame kernel (vfa_pipe.hip) walks its share of the frame,
// as plain host / device code: tests/pipe_seq_harness.cpp compiles it with g++ and checks it on the CPU.
//
// Work of a frame, for ANY number of z-layers (reference vfa/model/vfa_op.py:50-59, 118-125: K = nl * C):
//   tile  = 8 x 4 BEV cells = one 32-row block of the matrix pipe
//   group = up to four live views of one (tile, scale): their accumulators (4 x 32 rows x 256 columns, fp32) stay in
//           registers over all layers, because `relu` follows the sum over the WHOLE K = nl * 256 (vfa_op.py:123-124)
//   phase = (group, layer)
//   step  = (phase, channel quarter q, set): set 0 = sub-tiles 0, 1 of the group, set 1 = sub-tiles 2, 3; one step is 64 rows x
//           64 channels of voxel features, pooled by the four pooling waves while the eight matrix waves multiply the
//           previous step; both sets of a (layer, quarter) use the same 64 x 256 slice of `collapse.weight`.
//           EVERY (phase, quarter) has both steps, also when the group has no third view (the step is then empty: a barrier
//           and nothing else): the set of a step is the parity of its index, so the kernel's loop, unrolled by two, addresses
//           the accumulators of each set statically.
// A workgroup owns the groups from the k_begin-th group of tile t_begin up to (not including) the k_end-th group of tile
// t_end (groups of a tile in (scale, view) order).
#ifndef VFA_PIPE_SEQ_H
#define VFA_PIPE_SEQ_H

#if defined(__HIPCC__)
#define VFA_SEQ_HD __host__ __device__ __forceinline__
#else
#define VFA_SEQ_HD inline
#endif

namespace vfa_pipe {

constexpr int kSeqMaxScales = 3;
constexpr int kGroupViews = 4;

VFA_SEQ_HD int seq_popc(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}
VFA_SEQ_HD int seq_ctz(unsigned v) { return __builtin_ctz(v); }

// groups of a (tile, scale) whose live-view mask is `m`, and the sets (pairs of sub-tiles) of all of them
VFA_SEQ_HD int groups_of(unsigned m) { return (seq_popc(m) + kGroupViews - 1) / kGroupViews; }
VFA_SEQ_HD int sets_of(unsigned m) { return (seq_popc(m) + 1) / 2; }

struct Step {
    int tile;        // < 0: no step
    int scale, layer, q, set;
    int nj;          // sub-tiles (views) of the group, 1..4
    unsigned views;  // view of sub-tile j in bits 8 j .. 8 j + 7
    int phase;       // running number of the (group, layer) of this step within the workgroup
    int rank;        // index of the group among the groups of its tile
    int index;       // running number of the step within the workgroup (parity = LDS buffer)
    bool grp_first;  // first step of the group for this set: the accumulators of the set start from the bias
    bool grp_last;   // last step of the group for this set: relu and the view sum follow
    bool tile_last;  // last step of this workgroup's part of the tile
    VFA_SEQ_HD bool valid() const { return tile >= 0; }
    VFA_SEQ_HD int view(int j) const { return (int)((views >> (8 * j)) & 0xffu); }
    VFA_SEQ_HD int in_set() const { return nj - 2 * set >= 2 ? 2 : (nj - 2 * set > 0 ? nj - 2 * set : 0); } // sub-tiles of this step: 0, 1 or 2
    VFA_SEQ_HD bool same_chunk(const Step &o) const { return scale == o.scale && layer == o.layer && q == o.q; }
};

// A phase = (group, layer): eight steps, quarter q = k >> 1, set = k & 1 for k = 0 .. 7.  The kernel hands PHASES from the one
// wave that runs the generator to the others (through LDS) and every wave derives the steps itself (step_of).
struct Phase {
    int tile;        // < 0: no phase
    int scale, layer, nj;
    unsigned views;
    int phase, rank;
    bool more_in_tile; // another group of this workgroup follows in the same tile
    VFA_SEQ_HD bool valid() const { return tile >= 0; }
};

VFA_SEQ_HD Step step_of(const Phase &ph, int k, int nl, int index)
{
    Step st;
    st.tile = ph.tile; st.scale = ph.scale; st.layer = ph.layer; st.q = k >> 1; st.set = k & 1; st.nj = ph.nj; st.views = ph.views;
    st.phase = ph.phase; st.rank = ph.rank; st.index = index;
    st.grp_first = ph.layer == 0 && st.q == 0;
    st.grp_last = ph.layer == nl - 1 && st.q == 3;
    st.tile_last = st.grp_last && st.set == 1 && !ph.more_in_tile;
    return st;
}

// Generator of the phases / steps of one workgroup.  `Masks` returns the live-view mask of (scale, tile).
template <class Masks>
struct Sequencer {
    Masks masks;
    int n_scales, nl, t_end, k_end, t_lim;
    // position
    int tile, scale, rank, layer, q, set, nj, phase, index;
    unsigned rest, views, m0, m1, m2; // (scalars, not an array: a dynamically indexed array would live in scratch memory)
    bool in_group, more_in_tile;

    // (three value selects: `s == 0 ? m0 : ...` becomes a load through a selected ADDRESS and keeps the whole object in memory)
    VFA_SEQ_HD unsigned mask_of(int s) const { return (s == 0 ? m0 : 0u) | (s == 1 ? m1 : 0u) | (s == 2 ? m2 : 0u); }
    VFA_SEQ_HD void load_tile()
    {
        const bool on = tile < t_lim;
        m0 = on ? masks(0, tile) : 0u;
        m1 = (on && n_scales > 1) ? masks(1, tile) : 0u;
        m2 = (on && n_scales > 2) ? masks(2, tile) : 0u;
    }
    VFA_SEQ_HD void begin(int n_scales_, int nl_, int t_begin, int k_begin, int t_end_, int k_end_)
    {
        n_scales = n_scales_; nl = nl_; t_end = t_end_; k_end = k_end_;
        t_lim = k_end > 0 ? t_end + 1 : t_end;
        tile = t_begin; scale = 0; rank = 0; layer = q = set = 0; nj = 0; phase = -1; index = -1; cur.tile = -1;
        views = 0; in_group = false; more_in_tile = false;
        load_tile();
        rest = m0;
        // the first k_begin groups of the tile belong to the workgroup in front
        for (int skip = k_begin; skip > 0 && tile < t_lim;) {
            if (rest) {
                for (int j = 0; j < kGroupViews && rest; ++j) rest &= rest - 1u;
                --skip; ++rank;
            } else if (scale + 1 < n_scales) rest = mask_of(++scale);
            else break;
        }
    }
    // true when a group was formed
    VFA_SEQ_HD bool next_group()
    {
        while (tile < t_lim) {
            if (rest) {
                if (tile == t_end && rank >= k_end) return false; // the next workgroup's part of the tile
                views = 0; nj = 0;
                for (; nj < kGroupViews && rest; ++nj) {
                    views |= (unsigned)seq_ctz(rest) << (8 * nj);
                    rest &= rest - 1u;
                }
                bool later = rest != 0u;
                if (scale < 1 && n_scales > 1) later = later || m1 != 0u;
                if (scale < 2 && n_scales > 2) later = later || m2 != 0u;
                more_in_tile = later && !(tile == t_end && rank + 1 >= k_end);
                return true;
            }
            if (scale + 1 < n_scales) rest = mask_of(++scale);
            else {
                ++tile; scale = 0; rank = 0;
                load_tile();
                rest = m0;
            }
        }
        return false;
    }
    VFA_SEQ_HD Phase next_phase()
    {
        Phase ph;
        ph.tile = -1; ph.scale = ph.layer = ph.nj = 0; ph.views = 0; ph.phase = ph.rank = 0; ph.more_in_tile = false;
        if (in_group) {
            if (layer + 1 < nl) ++layer;
            else { in_group = false; ++rank; }
        }
        if (!in_group) {
            if (!next_group()) return ph;
            in_group = true;
            layer = 0;
        }
        ++phase;
        ph.tile = tile; ph.scale = scale; ph.layer = layer; ph.nj = nj; ph.views = views; ph.phase = phase; ph.rank = rank;
        ph.more_in_tile = more_in_tile;
        return ph;
    }
    // the steps one by one (CPU harness; the kernel expands the phases itself)
    Phase cur;
    VFA_SEQ_HD Step next()
    {
        if (index < 0 || q == 7) { cur = next_phase(); q = 0; }
        else ++q; // (q doubles as the step counter inside the phase here)
        if (!cur.valid()) { Step st; st.tile = -1; st.scale = st.layer = st.q = st.set = st.nj = 0; st.views = 0; st.phase = st.rank = st.index = 0;
                            st.grp_first = st.grp_last = st.tile_last = false; q = 7; index = index < 0 ? 0 : index; return st; }
        ++index;
        return step_of(cur, q, nl, index);
    }
};

// ------------------------------------------------------------------------------------------------
// work cuts: the groups of the frame in (tile, scale, view) order, cut into `n_chunks` pieces of equal estimated cost.
// Cost model in units of 64 cycles, fitted to the per-workgroup cycle counts of the diagnostic build of the sixteen-wave kernel on
// the bench frame and three multi-layer frames (tools/bench_pipe.py --fit): a step with work 69, a step of an empty set 46 (it
// still waits for the next step's window), a sub-tile pooled from L2 instead of an LDS window + 35 per step, + 14 per group
// (weight reloads, the running sum), + 36 per tile (its store).
// ------------------------------------------------------------------------------------------------
constexpr unsigned kStepCost = 69, kEmptyStepCost = 46, kGlobStepCost = 35, kGroupCost = 14, kTileCost = 36, kEmptyCost = 1;

VFA_SEQ_HD unsigned group_cost(int nj, int nl)
{
    const unsigned sets = (unsigned)((nj + 1) >> 1);
    return 4u * (unsigned)nl * (kStepCost * sets + kEmptyStepCost * (2u - sets)) + kGroupCost;
}

// walks the groups of a tile (masks m[0 .. n_scales)): visit(k, w0, w1) for the k-th group covering [w0, w1) of the tile's cost;
// `globs` = (view, layer, scale) items of the tile that are pooled from L2 (charged to the first group); returns the tile's cost
template <class Visit>
VFA_SEQ_HD unsigned walk_tile(const unsigned *m, int n_scales, int nl, unsigned globs, Visit &&visit)
{
    unsigned w = 0;
    int k = 0;
    for (int s = 0; s < n_scales; ++s) {
        int left = seq_popc(m[s]);
        while (left > 0) {
            const int nj = left < kGroupViews ? left : kGroupViews;
            unsigned wi = group_cost(nj, nl);
            if (k == 0) wi += kTileCost + 4u * kGlobStepCost * globs;
            visit(k, w, w + wi);
            w += wi;
            ++k;
            left -= nj;
        }
    }
    return k == 0 ? kEmptyCost : w;
}

} // namespace vfa_pipe
#endif // VFA_PIPE_SEQ_H
