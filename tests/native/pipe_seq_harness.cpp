// CPU check of vfa_amd/csrc/vfa_pipe_seq.h (the step order of the pipelined frame kernel): for random live-view masks and
// random work cuts every (tile, scale, live view, layer, quarter) is visited exactly once, in groups of <= 4 sub-tiles of ONE run and
// ONE scale taken in (tile, view) order across the tiles of the run, with consistent first / last flags; the group count of the
// cost walk (walk_run) equals the generator's (walk_groups).  Built and run by tests/test_pipe_seq.py.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
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

// serial restatement of pipe_cuts_kernel: chunk c starts at the group (start[c], rank[c]) -- start in RUNS; `subcost[s][tile * n_views + view]`
// = the sub-tile's cost over its layers (what pipe_records_kernel adds up)
static void serial_cuts(const std::vector<unsigned> &live, const std::vector<unsigned> &subcost, int n_tiles, int n_views, int n_scales, int nl, int rt,
                        int n_chunks, std::vector<int> &start, std::vector<int> &rank)
{
    const int n_runs = runs_of(n_tiles, rt);
    auto mask_of = [&](int r) { return [&, r](int s, int off) { return r * rt + off < n_tiles ? live[(size_t)s * n_tiles + r * rt + off] : 0u; }; };
    auto cost_of = [&](int r) { return [&, r](int s, int off, int v) { return subcost[((size_t)s * n_tiles + r * rt + off) * n_views + v]; }; };
    auto tiles_of = [&](int r) { return n_tiles - r * rt < rt ? n_tiles - r * rt : rt; };
    std::vector<unsigned long long> before(n_runs + 1, 0);
    for (int r = 0; r < n_runs; ++r) before[r + 1] = before[r] + walk_run(n_scales, rt, nl, tiles_of(r), mask_of(r), cost_of(r), [](int, unsigned, unsigned) {});
    const unsigned long long total = before[n_runs];
    start.assign(n_chunks + 1, n_runs);
    rank.assign(n_chunks + 1, 0);
    auto pos_of = [&](long long c) { return (total * (unsigned long long)c + n_chunks - 1) / n_chunks; };
    long long c = 0;
    for (int r = 0; r < n_runs; ++r) {
        const unsigned long long tb = before[r];
        const int n_groups = groups_of_run(n_scales, rt, mask_of(r));
        const unsigned w = walk_run(n_scales, rt, nl, tiles_of(r), mask_of(r), cost_of(r), [&](int kk, unsigned w0, unsigned w1) {
            while (c < n_chunks) {
                const unsigned long long pc = pos_of(c);
                if (pc >= tb + w1) break;
                const int k = ((unsigned)(pc - tb) - w0) * 2 < (w1 - w0) ? kk : kk + 1;
                if (k >= n_groups) { start[c] = r + 1 < n_runs ? r + 1 : n_runs; rank[c] = 0; }
                else { start[c] = r; rank[c] = k; }
                ++c;
            }
        });
        if (n_groups == 0)
            for (; c < n_chunks && pos_of(c) < tb + w; ++c) { start[c] = r; rank[c] = 0; }
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
        const int rts[3] = {1, 2, 4};
        const int rt = rts[rng() % 3];
        const int n_runs = runs_of(n_tiles, rt);
        const unsigned vm = (1u << n_views) - 1u;
        const unsigned density = rng() % 4;
        std::vector<unsigned> live((size_t)n_scales * n_tiles);
        for (auto &x : live) {
            x = rng() & vm;
            if (density == 0) x &= rng();
            if (density == 1 && rng() % 3 == 0) x = 0;
            if (density == 2) x = vm;
        }
        // a sub-tile's cost over its layers, as the geometry pass would add it up (any value: the cuts only weigh with it)
        std::vector<unsigned> subcost((size_t)n_scales * n_tiles * n_views);
        for (auto &x : subcost) x = (unsigned)nl * (188u + rng() % 1200u);
        // the cost walk, the mask-only count and the generator agree on the number of groups of every run, and a run's groups are
        // full except the last one of a scale
        for (int r = 0; r < n_runs; ++r) {
            auto mask = [&](int s, int off) { return r * rt + off < n_tiles ? live[(size_t)s * n_tiles + r * rt + off] : 0u; };
            int n_cost = 0;
            walk_run(n_scales, rt, nl, 1, mask, [&](int s, int off, int v) { return subcost[((size_t)s * n_tiles + r * rt + off) * n_views + v]; },
                     [&](int k, unsigned, unsigned) { n_cost = k + 1; });
            int last_scale = -1, last_nj = 4;
            const int n_gen = walk_groups(n_scales, rt, 0, mask,
                                          [&](int, int s, unsigned, int nj, int) { if (s == last_scale && last_nj != 4) std::abort(); last_scale = s; last_nj = nj; });
            CHECK(n_cost == n_gen && groups_of_run(n_scales, rt, mask) == n_gen);
            // the walk by scale (what the cuts kernel runs: a thread per (run, scale)) lists the same groups with the same cost intervals
            auto cost = [&](int s, int off, int v) { return subcost[((size_t)s * n_tiles + r * rt + off) * n_views + v]; };
            std::vector<std::pair<unsigned, unsigned>> by_run, by_scale;
            const int tiles = std::min(rt, n_tiles - r * rt);
            const unsigned w_run = walk_run(n_scales, rt, nl, tiles, mask, cost, [&](int, unsigned w0, unsigned w1) { by_run.push_back({w0, w1}); });
            unsigned w_scales = 0;
            bool first = true;
            for (int s = 0; s < n_scales; ++s) {
                const unsigned base = w_scales;
                const unsigned ws = walk_scale(rt, nl, tiles, s, first, mask, cost, [&](int, unsigned w0, unsigned w1) { by_scale.push_back({base + w0, base + w1}); });
                if (ws) first = false;
                w_scales += ws;
            }
            CHECK(by_run == by_scale);
            CHECK(n_gen == 0 ? w_run == kEmptyCost * (unsigned)tiles : w_run == w_scales);
        }
        std::vector<int> start, rank;
        serial_cuts(live, subcost, n_tiles, n_views, n_scales, nl, rt, n_chunks, start, rank);
        CHECK(start[0] == 0 && rank[0] == 0);
        CHECK(start[n_chunks] == n_runs && rank[n_chunks] == 0);
        // (tile, scale, view) -> count of (layer, q) visits
        std::map<std::tuple<int, int, int>, int> seen;
        std::map<int, int> run_last_count, run_parts;
        for (int wg = 0; wg < nblk; ++wg) {
            const int c0 = (int)((long long)n_chunks * wg / nblk), c1 = (int)((long long)n_chunks * (wg + 1) / nblk);
            const int rb = start[c0], kb = rank[c0], re = start[c1], ke = rank[c1];
            CHECK(rb < re || (rb == re && kb <= ke));
            if (rb > re || (rb == re && kb >= ke)) continue;
            Sequencer<HostMasks> sq;
            sq.masks = HostMasks{&live, n_tiles};
            sq.begin(n_scales, nl, n_tiles, rt, rb, kb, re, ke);
            Step prev; prev.run = -1;
            int steps = 0, cur_phase = -1;
            std::map<int, int> touched;
            // contributions of this workgroup: (tile, scale) -> indices stored so far (one per (group, tile): vfa_pipe_seq.h)
            std::map<std::pair<int, int>, int> contrib;
            int prev_group_phase = -1;
            for (;;) {
                const Step st = sq.next();
                if (!st.valid()) break;
                CHECK(st.index == steps);
                ++steps;
                CHECK(st.run >= rb && st.run < (ke > 0 ? re + 1 : re));
                CHECK(st.nj >= 1 && st.nj <= 4 && st.set == (st.index & 1));
                CHECK(st.layer >= 0 && st.layer < nl && st.q >= 0 && st.q < 4);
                CHECK(st.phase == cur_phase || st.phase == cur_phase + 1);
                cur_phase = st.phase;
                CHECK(st.grp_first == (st.layer == 0 && st.q == 0));
                CHECK(st.grp_last == (st.layer == nl - 1 && st.q == 3));
                if (prev.valid() && prev.run == st.run) CHECK(!prev.run_last);
                if (prev.valid() && prev.run != st.run) CHECK(prev.run_last);
                for (int j = 1; j < st.nj; ++j) // sub-tiles of a group in (tile, view) order
                    CHECK(st.tile(j, rt) > st.tile(j - 1, rt) || (st.tile(j, rt) == st.tile(j - 1, rt) && st.view(j) > st.view(j - 1)));
                for (int x = 0; x < st.in_set(); ++x) {
                    const int j = 2 * st.set + x, v = st.view(j), t = st.tile(j, rt);
                    CHECK(t < n_tiles && t / rt == st.run);
                    CHECK(v < n_views && ((live[(size_t)st.scale * n_tiles + t] >> v) & 1u));
                    seen[std::make_tuple(t, st.scale, v)] += 1;
                }
                if (st.grp_last && st.set == 1 && st.phase != prev_group_phase) { // the group ends: one contribution per tile of the group
                    prev_group_phase = st.phase;
                    CHECK(st.ci >= 0 && st.ci < contributions_of(n_views));
                    for (int j = 0; j < st.nj; ++j) {
                        if (j > 0 && st.tile(j, rt) == st.tile(j - 1, rt)) continue;
                        const int ci = j == 0 ? st.ci : 0;
                        int &cnt = contrib[std::make_pair(st.tile(j, rt), st.scale)];
                        CHECK(ci == cnt); // indices of a (tile, scale) are handed out consecutively from 0 inside a workgroup
                        ++cnt;
                        CHECK(cnt <= contributions_of(n_views));
                    }
                }
                if (st.run_last) run_last_count[st.run] += 1;
                touched[st.run] = 1;
                prev = st;
            }
            if (prev.valid()) CHECK(prev.run_last);
            for (auto &kv : touched) run_parts[kv.first] += 1;
        }
        for (int t = 0; t < n_tiles; ++t)
            for (int s = 0; s < n_scales; ++s)
                for (int v = 0; v < n_views; ++v) {
                    const bool on = (live[(size_t)s * n_tiles + t] >> v) & 1u;
                    auto it = seen.find(std::make_tuple(t, s, v));
                    CHECK((it == seen.end() ? 0 : it->second) == (on ? 4 * nl : 0));
                }
        for (auto &kv : run_parts) CHECK(run_last_count[kv.first] == kv.second); // every part of a run ends with a run_last step
    }
    std::printf("ok %d rounds\n", rounds);
    return 0;
}
