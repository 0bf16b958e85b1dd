// groups_test.cpp — host test of fusion-sim_amd/csrc/fes_groups.hpp (built by tests/test_groups.py with g++).
// Random bin tables of 1..4 species over tile layers, random piece sizes; the joint work list as joint_scan_kernel lays it out;
// then what the kernels do: every item takes groups [g_begin, g_end) of PPT slots of every species.  Checked:
//   * one launch (part 0): every group of every species is taken exactly once;
//   * two parts: again exactly once over both; part 2 (the interior) takes no group that holds a slot of the layers along
//     the faces or beyond (>= B, or below the interior's first slot ... the lower face's straddling group goes to part 1);
//   * the single-species form (chunk 0, items with explicit begin / end) gives the same;
//   * the exact rule of the full-EM push (groups_exact, owns): every slot once, by its own tile's item, no vector store over
//     slots of another item.
#include "../../fusion-sim_amd/csrc/fes_groups.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct Item { uint32_t tile, begin, end; };

static int fail(const char* what, int seed) { std::printf("FAILED: %s (seed %d)\n", what, seed); return 1; }

int main()
{
    int cases = 0;
    for (int seed = 1; seed <= 4000; ++seed) {
        std::mt19937 rng(seed);
        auto rnd = [&](int lo, int hi) { return lo + static_cast<int>(rng() % static_cast<unsigned>(hi - lo + 1)); };
        const int nsp = rnd(1, 4), ppt = rnd(0, 1) ? 4 : 2;
        const int per_layer = rnd(1, 5), layers = rnd(3, 9), ntiles = per_layer * layers;
        const uint32_t chunk = static_cast<uint32_t>(rnd(1, 40));
        const int lo = rnd(1, layers - 2), hi = rnd(lo + 1, layers - 1); // interior layers [lo, hi): at least one, faces on both sides
        std::vector<std::vector<uint32_t>> start(nsp, std::vector<uint32_t>(ntiles + 1, 0));
        for (int s = 0; s < nsp; ++s)
            for (int t = 0; t < ntiles; ++t) start[s][t + 1] = start[s][t] + (rnd(0, 3) == 0 ? 0u : static_cast<uint32_t>(rnd(0, 90)));
        // the joint list: tile t contributes max_s pieces
        std::vector<Item> joint;
        for (int t = 0; t < ntiles; ++t) {
            uint32_t m = 0;
            for (int s = 0; s < nsp; ++s) { const uint32_t k = fesgrp::pieces_of(start[s][t + 1] - start[s][t], chunk); m = k > m ? k : m; }
            for (uint32_t k = 0; k < m; ++k) joint.push_back({ static_cast<uint32_t>(t), k, 0 });
        }
        for (int s = 0; s < nsp; ++s) {
            const uint32_t n = start[s][ntiles], B = start[s][hi * per_layer], A = start[s][lo * per_layer];
            const size_t ngroups = (static_cast<size_t>(n) + ppt - 1) / ppt;
            // own list of the species (bin_scan_kernel's form): explicit ranges
            std::vector<Item> own;
            for (int t = 0; t < ntiles; ++t)
                for (uint32_t b = start[s][t]; b < start[s][t + 1]; b += chunk) own.push_back({ static_cast<uint32_t>(t), b, std::min(b + chunk, start[s][t + 1]) });
            for (int form = 0; form < 2; ++form) {
                const std::vector<Item>& list = form ? own : joint;
                for (int two_parts = 0; two_parts < 2; ++two_parts) {
                    std::vector<int> taken(ngroups + 1, 0), by_part2(ngroups + 1, 0);
                    for (int part = two_parts ? 1 : 0; part <= (two_parts ? 2 : 0); ++part) {
                        for (const Item& w : list) {
                            const int layer = static_cast<int>(w.tile) / per_layer;
                            const bool interior = layer >= lo && layer < hi;
                            if (part != 0 && (part == 2) != interior) continue; // in_part
                            uint32_t b0 = w.begin, b1 = w.end;
                            if (!form) fesgrp::piece(start[s][w.tile], start[s][w.tile + 1], w.begin, chunk, b0, b1);
                            size_t g0, g1;
                            fesgrp::groups(b0, b1, ppt, part, part ? A : 0u, part ? B : 0u, n, g0, g1);
                            for (size_t g = g0; g < g1; ++g) {
                                if (g >= ngroups) return fail("a group beyond the array", seed);
                                taken[g]++;
                                if (part == 2) by_part2[g] = 1;
                            }
                        }
                    }
                    for (size_t g = 0; g < ngroups; ++g) {
                        if (taken[g] != 1) return fail(taken[g] ? "a group taken twice" : "a group nobody took", seed);
                        // what the interior's launch pushed lies wholly inside the interior's slots, except for the one
                        // group that straddles the LOWER face's last slots (it belongs to the earlier item: part 1)
                        if (by_part2[g]) {
                            const size_t first = g * ppt, last = std::min<size_t>(n, first + ppt) - 1;
                            if (last >= B) return fail("the interior's launch pushed a particle of the upper face layer", seed);
                            if (first < A) return fail("the interior's launch pushed a particle of the lower face layer", seed);
                        }
                    }
                    ++cases;
                }
                // the exact rule (groups_exact / owns: the full-EM push): every slot is pushed once, by the item whose range
                // holds it — whatever the part — and a group is stored as one vector only by an item that owns all of it
                {
                    std::vector<int> pushed(n, 0), vector_stores(ngroups + 1, 0), scalar_touch(ngroups + 1, 0);
                    for (const Item& w : list) {
                        uint32_t b0 = w.begin, b1 = w.end;
                        if (!form) fesgrp::piece(start[s][w.tile], start[s][w.tile + 1], w.begin, chunk, b0, b1);
                        size_t g0, g1;
                        fesgrp::groups_exact(b0, b1, ppt, g0, g1);
                        if (b1 > b0 && (g0 * ppt > b0 || g1 * ppt < b1)) return fail("exact: the groups do not cover the item", seed);
                        if (b1 == b0 && g1 != g0) return fail("exact: an empty item has groups", seed);
                        for (size_t g = g0; g < g1; ++g) {
                            const size_t base = g * ppt;
                            const bool whole = base >= b0 && base + ppt <= b1;
                            if (whole) vector_stores[g]++; else scalar_touch[g]++;
                            for (int q = 0; q < ppt; ++q)
                                if (whole || fesgrp::owns(b0, b1, base + q)) {
                                    if (base + q >= n) return fail("exact: a slot beyond the array", seed);
                                    if (base + q < start[s][w.tile] || base + q >= start[s][w.tile + 1]) return fail("exact: a slot of another tile", seed);
                                    pushed[base + q]++;
                                }
                        }
                    }
                    for (uint32_t i = 0; i < n; ++i) if (pushed[i] != 1) return fail(pushed[i] ? "exact: a slot pushed twice" : "exact: a slot nobody pushed", seed);
                    for (size_t g = 0; g < ngroups; ++g) if (vector_stores[g] > 1 || (vector_stores[g] && scalar_touch[g])) return fail("exact: a vector store over another item's slots", seed);
                    ++cases;
                }
            }
        }
    }
    // held_plane: a rank's slab with `halo` planes on either side (+ 1: nodes), periodic; every plane the rank touches has
    // its own place in [0, nzs), every other plane none — for every rank, the first and last included (their halo wraps)
    int planes = 0;
    for (int seed = 1; seed <= 2000; ++seed) {
        std::mt19937 rng(seed);
        auto rnd = [&](int lo, int hi) { return lo + static_cast<int>(rng() % static_cast<unsigned>(hi - lo + 1)); };
        const int world = rnd(2, 9), nzl = rnd(3, 40), nz = world * nzl, halo = rnd(1, 8);
        const int nzs = nzl + 2 * halo + 1;
        if (nzs >= nz) continue; // (such a rank keeps whole-grid arrays)
        for (int rank = 0; rank < world; ++rank) {
            const int z0 = rank * nzl;
            const fes::Held hd{ ((z0 - halo) % nz + nz) % nz, nzs };
            std::vector<int> seen(nzs, 0);
            for (int k = 0; k < nz; ++k) {
                int d = k - (z0 - halo);                 // distance above the first held plane, periodic
                d = (d % nz + nz) % nz;
                const int l = fes::held_plane(k, hd, nz);
                if (d < nzs) { if (l != d) return fail("held_plane: a held plane in the wrong place", seed); seen[l]++; }
                else if (l != -1) return fail("held_plane: a plane the rank does not hold has a place", seed);
                ++planes;
            }
            for (int v : seen) if (v != 1) return fail("held_plane: places not one to one", seed);
        }
    }
    // wrap_near(d, n) == d mod n for -n < d < 2 n
    for (int n = 1; n <= 70; ++n)
        for (int d = -n + 1; d < 2 * n; ++d)
            if (fes::wrap_near(d, n) != static_cast<unsigned>(((d % n) + n) % n)) return fail("wrap_near", n);
    // wrap_window(d, n) == d mod n over everything a window can ask for: -n - 2 <= d < n + 32
    for (int n = 1; n <= 200; ++n)
        for (int d = -n - 2; d < n + 32; ++d)
            if (fes::wrap_window(d, n) != ((d % n) + n) % n) return fail("wrap_window", n);
    std::printf("cases=%d planes=%d\nok\n", cases, planes);
    return 0;
}
