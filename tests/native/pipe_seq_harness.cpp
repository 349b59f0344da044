// CPU check of vfa_amd/csrc/vfa_pipe_seq.h (the step order of the pipelined frame kernel): for random live-view masks and
// random work cuts every (tile, scale, live view, layer, quarter) is visited exactly once, in groups of <= 4 views, with
// consistent first / last flags.  Built and run by tests/test_pipe_seq.py.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <tuple>
#include <vector>

#include "../../vfa_amd/csrc/vfa_pipe_seq.h"

using namespace vfa_pipe;

struct HostMasks {
    const std::vector<unsigned> *live; // [scale][tile]
    int n_tiles;
    unsigned operator()(int s, int t) const { return (*live)[(size_t)s * n_tiles + t]; }
};

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s (seed %u)\n", __FILE__, __LINE__, #c, seed); return 1; } } while (0)

// serial restatement of pipe_cuts_kernel: chunk c starts at the group (start[c], rank[c])
static void serial_cuts(const std::vector<unsigned> &live, int n_tiles, int n_scales, int nl, int n_chunks, std::vector<int> &start,
                        std::vector<int> &rank)
{
    std::vector<unsigned long long> before(n_tiles + 1, 0);
    for (int t = 0; t < n_tiles; ++t) {
        unsigned m[3] = {0, 0, 0};
        for (int s = 0; s < n_scales; ++s) m[s] = live[(size_t)s * n_tiles + t];
        before[t + 1] = before[t] + walk_tile(m, n_scales, nl, 0u, [](int, unsigned, unsigned) {});
    }
    const unsigned long long total = before[n_tiles];
    start.assign(n_chunks + 1, n_tiles);
    rank.assign(n_chunks + 1, 0);
    auto pos_of = [&](long long c) { return (total * (unsigned long long)c + n_chunks - 1) / n_chunks; };
    long long c = 0;
    for (int t = 0; t < n_tiles; ++t) {
        unsigned m[3] = {0, 0, 0};
        for (int s = 0; s < n_scales; ++s) m[s] = live[(size_t)s * n_tiles + t];
        const unsigned long long tb = before[t];
        int n_groups = 0;
        walk_tile(m, n_scales, nl, 0u, [&](int k, unsigned, unsigned) { n_groups = k + 1; });
        const unsigned w = walk_tile(m, n_scales, nl, 0u, [&](int kk, unsigned w0, unsigned w1) {
            while (c < n_chunks) {
                const unsigned long long pc = pos_of(c);
                if (pc >= tb + w1) break;
                const int k = ((unsigned)(pc - tb) - w0) * 2 < (w1 - w0) ? kk : kk + 1;
                if (k >= n_groups) { start[c] = t + 1 < n_tiles ? t + 1 : n_tiles; rank[c] = 0; }
                else { start[c] = t; rank[c] = k; }
                ++c;
            }
        });
        if (n_groups == 0)
            for (; c < n_chunks && pos_of(c) < tb + w; ++c) { start[c] = t; rank[c] = 0; }
    }
}

int main(int argc, char **argv)
{
    const unsigned seed0 = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 200;
    for (int round = 0; round < rounds; ++round) {
        const unsigned seed = seed0 * 1000u + (unsigned)round;
        std::mt19937 rng(seed);
        const int n_tiles = 1 + (int)(rng() % 40), n_scales = 1 + (int)(rng() % 3), nl = 1 + (int)(rng() % 5);
        const int n_views = 1 + (int)(rng() % 12), n_chunks = 64, nblk = 1 + (int)(rng() % 24);
        const unsigned vm = (1u << n_views) - 1u;
        const unsigned density = rng() % 4;
        std::vector<unsigned> live((size_t)n_scales * n_tiles);
        for (auto &x : live) {
            x = rng() & vm;
            if (density == 0) x &= rng();
            if (density == 1 && rng() % 3 == 0) x = 0;
            if (density == 2) x = vm;
        }
        std::vector<int> start, rank;
        serial_cuts(live, n_tiles, n_scales, nl, n_chunks, start, rank);
        CHECK(start[0] == 0 && rank[0] == 0);
        CHECK(start[n_chunks] == n_tiles && rank[n_chunks] == 0);
        // (tile, scale, view) -> count of (layer, q) visits
        std::map<std::tuple<int, int, int>, int> seen;
        std::map<int, int> tile_last_count, tile_parts;
        for (int wg = 0; wg < nblk; ++wg) {
            const int c0 = (int)((long long)n_chunks * wg / nblk), c1 = (int)((long long)n_chunks * (wg + 1) / nblk);
            const int tb = start[c0], kb = rank[c0], te = start[c1], ke = rank[c1];
            CHECK(tb < te || (tb == te && kb <= ke));
            if (tb > te || (tb == te && kb >= ke)) continue;
            Sequencer<HostMasks> sq;
            sq.masks = HostMasks{&live, n_tiles};
            sq.begin(n_scales, nl, tb, kb, te, ke);
            Step prev; prev.tile = -1;
            int steps = 0, cur_phase = -1;
            std::map<int, int> touched;
            for (;;) {
                const Step st = sq.next();
                if (!st.valid()) break;
                CHECK(st.index == steps);
                ++steps;
                CHECK(st.tile >= tb && st.tile < (ke > 0 ? te + 1 : te));
                CHECK(st.nj >= 1 && st.nj <= 4 && st.set == (st.index & 1));
                CHECK(st.layer >= 0 && st.layer < nl && st.q >= 0 && st.q < 4);
                CHECK(st.phase == cur_phase || st.phase == cur_phase + 1);
                cur_phase = st.phase;
                CHECK(st.grp_first == (st.layer == 0 && st.q == 0));
                CHECK(st.grp_last == (st.layer == nl - 1 && st.q == 3));
                if (prev.valid() && prev.tile == st.tile) CHECK(!prev.tile_last);
                if (prev.valid() && prev.tile != st.tile) CHECK(prev.tile_last);
                for (int x = 0; x < st.in_set(); ++x) {
                    const int v = st.view(2 * st.set + x);
                    CHECK(v < n_views && ((live[(size_t)st.scale * n_tiles + st.tile] >> v) & 1u));
                    seen[std::make_tuple(st.tile, st.scale, v)] += 1;
                }
                if (st.tile_last) tile_last_count[st.tile] += 1;
                touched[st.tile] = 1;
                prev = st;
            }
            if (prev.valid()) CHECK(prev.tile_last);
            for (auto &kv : touched) tile_parts[kv.first] += 1;
        }
        for (int t = 0; t < n_tiles; ++t)
            for (int s = 0; s < n_scales; ++s)
                for (int v = 0; v < n_views; ++v) {
                    const bool on = (live[(size_t)s * n_tiles + t] >> v) & 1u;
                    auto it = seen.find(std::make_tuple(t, s, v));
                    CHECK((it == seen.end() ? 0 : it->second) == (on ? 4 * nl : 0));
                }
        for (auto &kv : tile_parts) CHECK(tile_last_count[kv.first] == kv.second); // every part of a tile ends with a tile_last step
    }
    std::printf("ok %d rounds\n", rounds);
    return 0;
}
