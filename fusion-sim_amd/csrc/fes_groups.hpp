// fes_groups.hpp — the integer rules of the tiled pushes that are easy to get wrong at an edge: which slots of a species'
// tile-ordered particle array a work item takes and in which groups of PPT slots, the place of a plane in a rank's
// slab-only arrays, the periodic distance inside a tile's window.  Plain functions shared by the kernels
// (fes_kernels.hpp) and a host test (tests/native/groups_test.cpp, g++): the one bug of round 3 that cost days was a group
// of four slots pushed by the wrong part of a two-part launch.
#ifndef FES_GROUPS_HPP
#define FES_GROUPS_HPP
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define FESGRP_HD __host__ __device__ __forceinline__
#else
#define FESGRP_HD inline
#endif

namespace fesgrp {

// items a tile contributes to a joint work list: the most pieces of `chunk` slots any species needs for it
FESGRP_HD uint32_t pieces_of(uint32_t count, uint32_t chunk) { return (count + chunk - 1) / chunk; }

// piece k of a tile whose slots are [t0, t1): [b0, b1), empty (b0 == b1 == t1) when the species has fewer pieces
FESGRP_HD void piece(uint32_t t0, uint32_t t1, uint32_t k, uint32_t chunk, uint32_t& b0, uint32_t& b1)
{
    const unsigned long long lo = static_cast<unsigned long long>(t0) + static_cast<unsigned long long>(k) * chunk;
    b0 = lo < t1 ? static_cast<uint32_t>(lo) : t1;
    b1 = t1 - b0 > chunk ? b0 + chunk : t1;
}

// The slots [b0, b1) of a work item in groups of ppt: groups [g_begin, g_end).  A group that straddles two items belongs
// to the earlier one (begin rounds up, end rounds up).  In a two-part launch (part 1: the tile layers along the slab's
// faces, part 2: the interior, whose slots are [A, B): B = the first slot of the layer along the upper face) the group
// that holds slot B holds particles of that face layer, which may deposit on planes that are exchanged before the
// interior is pushed: the interior gives it up (part 2) and the item that begins at B takes it (part 1) — when there is
// such an item (B < n: the species has particles at or beyond the upper face) and when the group's first slot is the
// interior's (>= A; with fewer interior slots than that the group begins in the lower face, whose item part 1 runs anyway).
// part 0: one launch, no exception.  n: the slots of the array.
FESGRP_HD void groups(uint32_t b0, uint32_t b1, int ppt, int part, uint32_t A, uint32_t B, uint32_t n, size_t& g_begin, size_t& g_end)
{
    g_begin = (static_cast<size_t>(b0) + ppt - 1) / ppt;
    g_end = (static_cast<size_t>(b1) + ppt - 1) / ppt;
    if (part == 2) { if (B < n && g_end > B / ppt) g_end = B / ppt; }
    else if (part == 1 && b0 == B && b1 > b0 && static_cast<size_t>(B / ppt) * ppt >= A) g_begin = B / ppt;
}

// The other rule (the full-EM push since round 4): a work item takes exactly its own slots.  Groups [g_begin, g_end) are those
// that hold any of [b0, b1); slot g * ppt + q of a group is the item's iff owns(b0, b1, slot); a group that is not wholly
// the item's is stored slot by slot, so two items that share a group never write each other's slots (what either reads of
// the other's it does not use).  Why: a shared group's foreign particles lie in ANOTHER tile, i.e. outside the window of the
// workgroup that pushes them, and take the global-memory path — up to ppt - 1 of them one after the other in one lane, at
// the end of every tile: measured 40 ns of launch time per tile in single precision (profiles/r04_em_tile_overhead.txt).
// No exception for two-part launches: an item belongs to one part with all its slots.
FESGRP_HD void groups_exact(uint32_t b0, uint32_t b1, int ppt, size_t& g_begin, size_t& g_end)
{
    g_begin = static_cast<size_t>(b0) / ppt;
    g_end = b1 > b0 ? (static_cast<size_t>(b1) + ppt - 1) / ppt : g_begin;
}
FESGRP_HD bool owns(uint32_t b0, uint32_t b1, size_t slot) { return slot >= b0 && slot < b1; }

} // namespace fesgrp

namespace fes {

// The node arrays of a handle hold the planes zs0, zs0 + 1, ..., zs0 + nzs - 1 along z (periodic in nz): all of them
// (zs0 = 0, nzs = nz) for an undecomposed handle or a rank that keeps global arrays; its slab with the ghost / halo planes
// for a rank of a COMPACT decomposition (fes_api.hip, keep_slab_only), whose arrays have nzs planes only.  Kernels take the
// global plane index of a node through held_plane(): the plane's place in the array, or -1 for a plane the handle does not
// hold (what a particle that has outrun the ghost planes would touch there is dropped: it was never read either).
struct Held {
    int zs0, nzs; // 0 <= zs0 < nz
};
FESGRP_HD int held_plane(int k, Held hd, int nz)
{
    int l = k - hd.zs0;
    if (l < 0) l += nz;
    return l < hd.nzs ? l : -1;
}

// d mod n for -n < d < 2 n, as an unsigned number: whichever of d, d + n, d - n lies in [0, n) is the smallest of the
// three taken as unsigned (one v_min3_u32 instead of two compare-and-select pairs)
FESGRP_HD unsigned wrap_near(int d, int n)
{
    const unsigned u = static_cast<unsigned>(d), m = static_cast<unsigned>(n);
    const unsigned a = u < u + m ? u : u + m, b = u - m;
    return a < b ? a : b;
}

// d mod n for the nodes of a tile's window: d = first node of the window (one or two halo cells below a tile, less the
// first plane a rank holds: down to -n - 2) + node within the window (< 32).  Boxes of 32 nodes or more along the axis
// take three compare-and-adds (-2 n <= d < 2 n), smaller ones the division (the choice is uniform over the launch).
FESGRP_HD int wrap_window(int d, int n)
{
    if (n < 32) { d %= n; return d < 0 ? d + n : d; }
    if (d < 0) d += n;
    if (d < 0) d += n;
    return d >= n ? d - n : d;
}

} // namespace fes
#endif
