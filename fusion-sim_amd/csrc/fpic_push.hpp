// fpic_push.hpp — the particle push: out.step (empic.js:1436-1469) as one kernel.
//
// One launch takes nsub = 2*ncalls leap-frog sub-steps.  A sub-step of one particle is
// step_velocity_frag (empic.js:749-773), then step_position_frag on the NEW velocity
// (empic.js:714-719), both reading the OLD random state, then the random state's own
// advance (empic.js:800-807) — the order of bindings in out.step (empic.js:815-853,
// :890-928).  The particle's state stays in registers between sub-steps, so the
// traffic per launch is one read and one write of 10 T + 1 byte per particle however
// many sub-steps are taken.
//
// Expression order follows the shader text (-ffp-contract=off, correctly rounded sqrt
// and divide): float results are bit-identical to the CPU oracle.  The radius and
// the nearest cell found by the boundary test of one sub-step ARE the radius and cell
// the next sub-step's velocity pass would recompute from the same position, so they
// are carried over instead (one sqrt and two NGP evaluations per sub-step instead
// of two and four).
//
// Three things ride on the tiled form of the kernel because the workgroup already owns
// a tile's particles and holds their final state in registers:
//   * the scatter's stage 1 (per-cell sums of density()'s point sprites),
//   * a census of the particles per tile (input of the next re-binning),
//   * the re-binning itself: on a re-binning launch the state is written to its
//     sorted place in the other particle set instead of in place.
#pragma once

#include "fpic_kernels.hpp"

namespace fpic {

template <typename T>
struct PushArgs {
    T* slab;               // 10 arrays x,y,z,vx,vy,vz,u1,u2,c1,c2, each `stride` elements
    size_t stride;
    uint8_t* alive;
    const T* coef;
    const uint8_t* sink_alive;
    const T* inv_cdf_xy;
    const T* entropy;
    int nr, nz;
    T step_factor;
    unsigned long long n;
    int nsub;              // two sub-steps per step() call; fpic_substeps() may pass any count
    // counter-based RNG extension (CTR kernels): nothing random is stored per particle
    const uint32_t* id;    // the caller's particle index = the generator's stream
    uint32_t seed_lo, seed_hi;
    unsigned long long t0; // global index of this launch's first sub-step
    int raster_bits;       // spec.raster_subpixel_bits: 0 = ideal point sprites, b = drawn as a rasteriser with b sub-pixel bits does
};

// Philox4x32-10 (Salmon et al., SC'11; Random123 constants).  Extension mode only:
// the random vector of particle `id` at sub-step `t` is the block with counter
// (id, t_lo, t_hi, 0x5EED) under key = seed, each word mapped to (w >> 8) * 2^-24.
// It stands where the reference reads its rand texel (empic.js:717, :772); the
// reference's own generator is the entropy-table walk of K3.
template <typename T>
__device__ __forceinline__ void counter_rand(uint32_t id, unsigned long long t, uint32_t k0, uint32_t k1, T (&u)[4])
{
    uint32_t c0 = id, c1 = static_cast<uint32_t>(t), c2 = static_cast<uint32_t>(t >> 32), c3 = 0x5EEDu;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0, n1 = static_cast<uint32_t>(p1);
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1, n3 = static_cast<uint32_t>(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u[0] = static_cast<T>(c0 >> 8) * static_cast<T>(1.0 / 16777216.0);
    u[1] = static_cast<T>(c1 >> 8) * static_cast<T>(1.0 / 16777216.0);
    u[2] = static_cast<T>(c2 >> 8) * static_cast<T>(1.0 / 16777216.0);
    u[3] = static_cast<T>(c3 >> 8) * static_cast<T>(1.0 / 16777216.0);
}

// Extra arguments of the tiled form.
template <typename T>
struct TileArgs {
    int ntx, ntz;
    uint32_t ntiles;                   // real tiles + 1 bin for clipped particles
    const BlockWork* work;             // chunks of the CURRENT particle order
    const uint32_t* nwork;
    T* cell_sums;                      // FUSE: sums_cells(nr, nz) x 4 (fpic_internal.hpp), zeroed by the host
    unsigned long long* spilled;       // FUSE: particles summed outside their LDS window
    uint32_t* tile_count;              // FUSE: census of final states per tile, zeroed by the host
    // SCATTER (re-binning launch): the other particle set and its bin table
    const uint32_t* id;
    T* dst_slab;
    uint8_t* dst_alive;
    uint32_t* dst_id;
    const uint32_t* dst_tile_start;
    uint32_t* dst_tile_cursor;
    // kNbrSlots counts per work item: written by an in-place fused launch (the census of its new positions per neighbour
    // tile), read by the re-binning launch that follows it when census_valid (then its count pass — one more read of x, y,
    // z — is skipped: the electrostatic push has done so since round 3)
    uint32_t* chunk_census;
    int census_valid;
};

template <typename T>
struct Particle {
    T x, y, z, vx, vy, vz, u1, u2, c1, c2;
    T r;        // sqrt(x*x + y*y) of the current position
    int ci, cj; // NGP cell of (r, z)
    bool alive;
    uint32_t at; // index of the particle in the arrays (CTR: where its id is found when needed)
};

// Where a sub-step reads the per-cell tables from.  GlobalTables: straight from
// L2/HBM (any particle order).  WindowTables: the tile's window staged in LDS by the
// workgroup, with global memory behind it for a particle outside the window.
template <typename T>
struct GlobalTables {
    const T* coef;
    const uint8_t* sink_alive;
    int nr;
    __device__ __forceinline__ void coefficients(int ci, int cj, T (&R1)[4], T (&R2)[4], T (&R3)[4]) const
    {
        const T* cf = coef + 12u * (static_cast<unsigned>(ci) + static_cast<unsigned>(nr) * static_cast<unsigned>(cj));
        load4(cf, R1);
        load4(cf + 4, R2);
        load4(cf + 8, R3);
    }
    __device__ __forceinline__ bool keep(int ci, int cj) const
    {
        return sink_alive[static_cast<unsigned>(ci) + static_cast<unsigned>(nr) * static_cast<unsigned>(cj)] != 0;
    }
};

constexpr int kPushHalo = 4;                           // cells staged around a tile for the push
constexpr int kPushLds = kTileSide + 2 * kPushHalo;    // 40
// Workgroup size of the tiled push (one workgroup per CU: 152 KB of LDS).  The reference generator's form holds 150 VGPRs —
// three waves per SIMD — and is bound by the table gather's misses: 768 threads (12 waves) beat 512 by 3 %, 1024 would spill.
// The counter-based form streams at 0.67 of the HBM peak and loses 11 % with 768.  (profiles/r04_push_threads.txt;
// FPIC_PUSH_THREADS / FPIC_PUSH_THREADS_CTR: development switches)
#if !defined(FPIC_PUSH_THREADS)
#define FPIC_PUSH_THREADS 768
#endif
#if !defined(FPIC_PUSH_THREADS_CTR)
#define FPIC_PUSH_THREADS_CTR 512
#endif
template <bool CTR>
constexpr int push_threads() { return CTR ? FPIC_PUSH_THREADS_CTR : FPIC_PUSH_THREADS; }
constexpr int kNbr = 5;                                // tile neighbourhood tracked in LDS when binning
constexpr int kNbrSlots = kNbr * kNbr + 1;             // + the bin of clipped particles
constexpr int kOwnSlot = (kNbr / 2) * kNbr + kNbr / 2; // the workgroup's own tile

// LDS pointers carry their address space in the type: through a generic pointer the
// compiler emits flat_load instead of ds_read_b128.  (HIP's float4/double2 classes
// cannot live in an address space; clang's native vectors can.)
#define FPIC_LDS __attribute__((address_space(3)))
template <typename T>
__device__ __forceinline__ void load4_lds(const FPIC_LDS T* p, T (&o)[4])
{
    using V = typename NatVec16<T>::type;
    if constexpr (sizeof(T) == 4) {
        const V v = *reinterpret_cast<const FPIC_LDS V*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        const V a = *reinterpret_cast<const FPIC_LDS V*>(p);
        const V b = *reinterpret_cast<const FPIC_LDS V*>(p + 2);
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    }
}

template <typename T>
struct WindowTables {
    GlobalTables<T> g;
    const FPIC_LDS T* lcoef;          // [kPushLds*kPushLds][12]
    const FPIC_LDS uint8_t* lsink;    // [kPushLds*kPushLds]
    int i0, j0;
    // The LDS read is unconditional (index 0 stands in for a cell outside the window);
    // only lanes outside the window, rare between binnings, run the global loads.
    __device__ __forceinline__ void coefficients(int ci, int cj, T (&R1)[4], T (&R2)[4], T (&R3)[4]) const
    {
        const unsigned li = static_cast<unsigned>(ci - i0), lj = static_cast<unsigned>(cj - j0);
        const bool in = li < static_cast<unsigned>(kPushLds) && lj < static_cast<unsigned>(kPushLds);
        const FPIC_LDS T* cf = lcoef + __umul24(12u, in ? __umul24(lj, kPushLds) + li : 0u); // (24-bit multiplies are full rate)
        load4_lds(cf, R1);
        load4_lds(cf + 4, R2);
        load4_lds(cf + 8, R3);
        if (!in) g.coefficients(ci, cj, R1, R2, R3);
    }
    __device__ __forceinline__ bool keep(int ci, int cj) const
    {
        const unsigned li = static_cast<unsigned>(ci - i0), lj = static_cast<unsigned>(cj - j0);
        const bool in = li < static_cast<unsigned>(kPushLds) && lj < static_cast<unsigned>(kPushLds);
        bool k = lsink[in ? __umul24(lj, kPushLds) + li : 0u] != 0;
        if (!in) k = g.keep(ci, cj);
        return k;
    }
};

template <typename T>
__device__ __forceinline__ void locate(Particle<T>& q, int nr, int nz)
{
    q.r = sqrt_(q.x * q.x + q.y * q.y);
    q.ci = ngp(q.r, nr);
    q.cj = ngp(q.z, nz);
}

// CTR = false: the reference's generator (per-particle state u1,u2,c1,c2 advanced by K3).
// CTR = true: the counter-based extension; tsub is the global index of this sub-step.
template <typename T, bool CTR, typename Tables>
__device__ __forceinline__ void substep(Particle<T>& q, const PushArgs<T>& a, const Tables& tab, unsigned long long tsub)
{
    T s[4] = { 0, 0, 0, 0 };
    if constexpr (!CTR) {
        // K3's entropy texel depends on nothing below: issue it first
        const unsigned et = static_cast<unsigned>(ngp(q.c1, kEntropySide)) + static_cast<unsigned>(kEntropySide) * static_cast<unsigned>(ngp(q.c2, kEntropySide));
        load4(a.entropy + 4u * et, s);
    }

    // K1: velocity in local cylindrical components, v' = R v + A at the nearest cell
    const T dx = q.x / q.r, dy = q.y / q.r;
    const T vr = q.vx * dx + q.vy * dy;
    const T va = q.vy * dx - q.vx * dy;
    T R1[4], R2[4], R3[4];
    tab.coefficients(q.ci, q.cj, R1, R2, R3);
    const T cx = ((R1[0] * vr + R1[1] * va) + R1[2] * q.vz) + R1[3];
    const T cy = ((R2[0] * vr + R2[1] * va) + R2[2] * q.vz) + R2[3];
    const T cz = ((R3[0] * vr + R3[1] * va) + R3[2] * q.vz) + R3[3];
    T nvx = cx * dx - cy * dy;
    T nvy = cx * dy + cy * dx;
    T nvz = cz;
    if (!q.alive) { // re-injected on the previous sub-step (empic.js:772, quirk Q4)
        T u[4] = { q.u1, q.u2, q.c1, q.c2 };
        if constexpr (CTR) counter_rand<T>(a.id[q.at], tsub, a.seed_lo, a.seed_hi, u);
        nvx = static_cast<T>(0.001) * (static_cast<T>(2) * u[0] - static_cast<T>(1));
        nvy = static_cast<T>(0.001) * (static_cast<T>(2) * u[1] - static_cast<T>(1));
        nvz = static_cast<T>(0.001) * (static_cast<T>(2) * u[2] - static_cast<T>(1));
    }
    q.vx = nvx; q.vy = nvy; q.vz = nvz;

    // K2: drift, boundary test at the new position's nearest cell
    q.x = q.x + a.step_factor * nvx;
    q.y = q.y + a.step_factor * nvy;
    q.z = q.z + a.step_factor * nvz;
    locate(q, a.nr, a.nz);
    q.alive = tab.keep(q.ci, q.cj);
    if (!q.alive) { // lost: re-inject from the inverse-CDF table on the plane y = 0 (empic.js:719)
        T u[4] = { q.u1, q.u2, q.c1, q.c2 };
        if constexpr (CTR) counter_rand<T>(a.id[q.at], tsub, a.seed_lo, a.seed_hi, u);
        const unsigned t = static_cast<unsigned>(ngp(u[0], kCdfSide)) + static_cast<unsigned>(kCdfSide) * static_cast<unsigned>(ngp(u[1], kCdfSide));
        q.x = a.inv_cdf_xy[2u * t];
        q.y = static_cast<T>(0);
        q.z = a.inv_cdf_xy[2u * t + 1];
        locate(q, a.nr, a.nz);
    }

    if constexpr (!CTR) {
        // K3: additive walk on (u1,u2), logistic map on (c1,c2) (quirk Q5: m == 1 stays 1)
        const T x0 = static_cast<T>(0.999) * q.c1 + static_cast<T>(0.001) * s[2];
        const T x1 = static_cast<T>(0.999) * q.c2 + static_cast<T>(0.001) * s[3];
        const T m0 = q.u1 + s[0], m1 = q.u2 + s[1];
        q.u1 = (m0 > static_cast<T>(1)) ? m0 - static_cast<T>(1) : m0;
        q.u2 = (m1 > static_cast<T>(1)) ? m1 - static_cast<T>(1) : m1;
        q.c1 = static_cast<T>(4) * x0 * (static_cast<T>(1) - x0);
        q.c2 = static_cast<T>(4) * x1 * (static_cast<T>(1) - x1);
    }
}

// Point-sprite cell of a state (deposit_cell on the carried radius): false = clipped.
template <typename T>
__device__ __forceinline__ bool sprite_cell(const Particle<T>& q, int nr, int nz, int& ic, int& jc)
{
    if (!(q.r >= static_cast<T>(0) && q.r <= static_cast<T>(1) && q.z >= static_cast<T>(0) && q.z <= static_cast<T>(1))) return false;
    ic = static_cast<int>(q.r * static_cast<T>(nr));
    jc = static_cast<int>(q.z * static_cast<T>(nz));
    return true;
}

// A workgroup tracks the 5x5 tiles around its own tile (+ the clipped bin) in LDS; a
// particle further away than that goes to the global tables directly.
struct TileNeighbourhood {
    int ti, tj, ntx, ntz;
    uint32_t clipped_bin;
    // slot in the LDS tables, or -1 for a tile outside the neighbourhood; key = global bin
    __device__ __forceinline__ int slot(bool visible, int ic, int jc, uint32_t& key) const
    {
        if (!visible) { key = clipped_bin; return kNbr * kNbr; }
        const int tx = ic / kTileSide, ty = jc / kTileSide;
        key = static_cast<uint32_t>(tx) + static_cast<uint32_t>(ntx) * static_cast<uint32_t>(ty);
        const unsigned dx = static_cast<unsigned>(tx - ti + kNbr / 2), dy = static_cast<unsigned>(ty - tj + kNbr / 2);
        return (dx < static_cast<unsigned>(kNbr) && dy < static_cast<unsigned>(kNbr)) ? static_cast<int>(dy * kNbr + dx) : -1;
    }
    // global bin of an LDS slot, or ~0u when the slot lies outside the tile grid
    __device__ __forceinline__ uint32_t bin_of_slot(int s) const
    {
        if (s == kNbr * kNbr) return clipped_bin;
        const int tx = ti + s % kNbr - kNbr / 2, ty = tj + s / kNbr - kNbr / 2;
        if (tx < 0 || tx >= ntx || ty < 0 || ty >= ntz) return ~0u;
        return static_cast<uint32_t>(tx) + static_cast<uint32_t>(ntx) * static_cast<uint32_t>(ty);
    }
};

// What happens to a particle's final state besides being stored.  NoSums: nothing.
// WindowSums: the scatter's stage 1 (programMoments01's vertex colour summed per
// nearest cell, see fpic_kernels.hpp) fused into the push: the workgroup owns the
// tile, the state is in registers and sqrt(x*x+y*y) has just been computed, so the
// separate pass that re-reads 24 B per particle disappears.  Accumulators are double
// in LDS (ds_add_f64; ds_add_f32 is 3.7x slower on gfx950).  It also counts the
// final states per tile: the census the next re-binning is laid out from.
struct NoSums {
    template <typename T>
    __device__ __forceinline__ void add(const Particle<T>&, int, int) const {}
};

template <typename T, bool SUMS = true>
struct WindowSums {
    FPIC_LDS double* lsums;      // [kTileLds*kTileLds][4]; unused when SUMS is false (census only)
    FPIC_LDS uint32_t* lcensus;  // [kNbrSlots]
    int i0, j0;                  // window origin (tile origin - kTileHalo)
    TileNeighbourhood nb;
    T* cell_sums;                // global sums_cells(nr, nz) x 4
    uint32_t* tile_count;        // global census
    unsigned* spilled;           // lane-local count of particles outside the window
    unsigned* own_census;        // lane-local count of final states in the workgroup's own tile
    int raster_bits;             // 0: ideal sprites; b: the sprite's cell as a rasteriser with b sub-pixel bits places it
    __device__ __forceinline__ void add(const Particle<T>& q, int nr, int nz) const
    {
        int ic = 0, jc = 0;
        bool visible = sprite_cell(q, nr, nz, ic, jc);   // the bins always follow the ideal cell
        uint32_t key;
        const int s = nb.slot(visible, ic, jc, key);
        // nearly every particle is still in the workgroup's own tile: those are counted in a register (64 lanes
        // adding to ONE LDS word are serialised), the others by LDS atomics
        if (s == kOwnSlot) ++*own_census;
        else if (s >= 0) __hip_atomic_fetch_add(lcensus + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(tile_count + key, 1u);
        if (!SUMS) return;
        const bool inside = visible;
        if (raster_bits) visible = raster_cell(q.r, q.z, nr, nz, raster_bits, ic, jc);   // uniform branch
        if (!visible) return;
        const T dx = q.x / q.r, dy = q.y / q.r;
        const T c0 = static_cast<T>(0.001) * (q.vx * dx + q.vy * dy);
        const T c1 = static_cast<T>(0.001) * (q.vy * dx - q.vx * dy);
        const T c2 = static_cast<T>(0.001) * q.vz;
        const unsigned li = static_cast<unsigned>(ic - i0), lj = static_cast<unsigned>(jc - j0);
        if (li < static_cast<unsigned>(kTileLds) && lj < static_cast<unsigned>(kTileLds)) {
            FPIC_LDS double* t = lsums + 4u * (__umul24(lj, kTileLds) + li);
            __hip_atomic_fetch_add(t, static_cast<double>(c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(t + 1, static_cast<double>(c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(t + 2, static_cast<double>(c2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // the count channel is an exact integer in the slot's low word (ds_add_u32 is the fast
            // LDS atomic); the flush turns n into n * 0.001 once
            __hip_atomic_fetch_add((FPIC_LDS uint32_t*)(t + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            T* g = cell_sums + 4 * sums_index(ic, jc, nr);
            atomicAdd(g, c0);
            atomicAdd(g + 1, c1);
            atomicAdd(g + 2, c2);
            atomicAdd(g + 3, static_cast<T>(0.001) * static_cast<T>(1));
            if (inside) ++*spilled;     // a point outside the unit square says nothing about the bins' age
        }
    }
};

// One lane owns the PPT consecutive particles starting at `base` (one 16-byte vector
// per array); cnt < PPT only for the last lane of the population.
template <typename T, bool CTR>
__device__ __forceinline__ void load_state(const PushArgs<T>& a, size_t base, int cnt, Particle<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    constexpr int NF = CTR ? 6 : 10; // the counter-based mode keeps no random state
    T v[10][PPT] = {};
    uint8_t al[PPT];
    // arrays are padded to a multiple of the vector width, so the vector load is in bounds
#pragma unroll
    for (int f = 0; f < NF; ++f) load_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
    if constexpr (PPT == 4) {
        const uchar4 b = *reinterpret_cast<const uchar4*>(a.alive + base);
        al[0] = b.x; al[1] = b.y; al[2] = b.z; al[3] = b.w;
    } else {
        const uchar2 b = *reinterpret_cast<const uchar2*>(a.alive + base);
        al[0] = b.x; al[1] = b.y;
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        q[k].x = v[0][k]; q[k].y = v[1][k]; q[k].z = v[2][k];
        q[k].vx = v[3][k]; q[k].vy = v[4][k]; q[k].vz = v[5][k];
        q[k].u1 = v[6][k]; q[k].u2 = v[7][k]; q[k].c1 = v[8][k]; q[k].c2 = v[9][k];
        q[k].alive = al[k] != 0;
        q[k].at = static_cast<uint32_t>(base + k);
        if (k >= cnt) { // padding lanes: keep every gather in range, results are discarded
            q[k].x = static_cast<T>(0.5); q[k].y = static_cast<T>(0); q[k].z = static_cast<T>(0.5);
            q[k].vx = q[k].vy = q[k].vz = static_cast<T>(0);
            q[k].u1 = q[k].u2 = q[k].c1 = q[k].c2 = static_cast<T>(0.5);
        }
        locate(q[k], a.nr, a.nz);
    }
}

template <typename T, bool CTR, typename Tables, typename Sums>
__device__ __forceinline__ void advance_state(const PushArgs<T>& a, const Tables& tab, const Sums& sums, int cnt,
                                              Particle<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    for (int s = 0; s < a.nsub; s += 2) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) substep<T, CTR>(q[k], a, tab, a.t0 + s);
        if (s + 1 < a.nsub) { // fpic_substeps() may ask for an odd number of sub-steps
#pragma unroll
            for (int k = 0; k < PPT; ++k) substep<T, CTR>(q[k], a, tab, a.t0 + s + 1);
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k)
        if (k < cnt) sums.add(q[k], a.nr, a.nz);
}

// in place, one 16-byte vector per array
template <typename T, bool CTR>
__device__ __forceinline__ void store_state(const PushArgs<T>& a, size_t base, int cnt, const Particle<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    constexpr int NF = CTR ? 6 : 10;
    T v[10][PPT];
    uint8_t al[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        v[0][k] = q[k].x; v[1][k] = q[k].y; v[2][k] = q[k].z;
        v[3][k] = q[k].vx; v[4][k] = q[k].vy; v[5][k] = q[k].vz;
        v[6][k] = q[k].u1; v[7][k] = q[k].u2; v[8][k] = q[k].c1; v[9][k] = q[k].c2;
        al[k] = q[k].alive ? 1 : 0;
    }
    if (cnt == PPT) {
#pragma unroll
        for (int f = 0; f < NF; ++f) store_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
        if constexpr (PPT == 4) {
            *reinterpret_cast<uchar4*>(a.alive + base) = make_uchar4(al[0], al[1], al[2], al[3]);
        } else {
            *reinterpret_cast<uchar2*>(a.alive + base) = make_uchar2(al[0], al[1]);
        }
    } else {
        // fully unrolled: a runtime index into v[][] would move the whole array to scratch
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (k < cnt) {
#pragma unroll
                for (int f = 0; f < NF; ++f) a.slab[f * a.stride + base + k] = v[f][k];
                a.alive[base + k] = al[k];
            }
        }
    }
}

// Flat form: any particle order, tables read from global memory.  Used until the
// particles have been binned.
template <typename T, bool CTR>
__global__ __launch_bounds__(256) void push_kernel(PushArgs<T> a)
{
    constexpr int PPT = Vec16<T>::N;
    const size_t base = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * PPT;
    if (base >= a.n) return;
    const int cnt = (base + PPT <= a.n) ? PPT : static_cast<int>(a.n - base);
    GlobalTables<T> tab{ a.coef, a.sink_alive, a.nr };
    Particle<T> q[PPT];
    load_state<T, CTR>(a, base, cnt, q);
    advance_state<T, CTR>(a, tab, NoSums{}, cnt, q);
    store_state<T, CTR>(a, base, cnt, q);
}

// Tiled form, for binned particles: one workgroup per chunk of one tile's particles
// (the binning's work list).  It first stages the tile's coefficient records and sink
// bytes, plus a 4-cell halo, in LDS; a lane then reads a particle's record with three
// ds_read_b128 instead of three divergent global loads (the L1 serves about one
// distinct line per clock per CU, which cost 1.4 ms of the flat kernel's 4.5 ms at
// 1e8 particles; profiles/r01_push_ablation.txt).  A particle that has drifted
// outside the window reads global memory; results never depend on the binning.
// Chunks are cut at arbitrary particle indices; a vector of PPT particles belongs
// to the chunk that holds its first particle.
//
// FUSE adds the scatter's stage 1 and the tile census (WindowSums): a second LDS
// window, tile + 8-cell halo of double accumulators, zeroed before and flushed with
// global float atomics in 256-byte pieces after the chunk.
//
// SCATTER (needs FUSE) makes this launch the re-binning as well.  The bin of a
// particle is the tile of the state it is LOADED with — exactly what the previous
// launch's census counted, from which the host has laid out dst_tile_start.  The
// workgroup first counts its chunk per bin, reserves one range per bin, then pushes
// and stores each final state at range + rank of arrival in the other particle set.
template <typename T>
constexpr size_t push_sums_offset() { return (static_cast<size_t>(kPushLds) * kPushLds * (12 * sizeof(T) + 1) + 15) / 16 * 16; }
template <typename T>
constexpr size_t push_stage_offset() // end of: coefficient window | sink bytes | double sums | census | ranks | ranges
{
    return (push_sums_offset<T>() + static_cast<size_t>(kTileLds) * kTileLds * 4 * sizeof(double) +
            3 * kNbrSlots * sizeof(uint32_t) + 15) / 16 * 16;
}
template <typename T, bool FUSE, bool SUMS = true>
constexpr size_t push_tiles_lds_bytes()
{
    // coefficient window | sink bytes | double sums window (SUMS) | census | ranks | ranges
    return !FUSE ? push_sums_offset<T>()
                 : SUMS ? push_stage_offset<T>() : push_sums_offset<T>() + 3 * kNbrSlots * sizeof(uint32_t) + 16;
}

template <typename T, bool FUSE, bool SCATTER, bool CTR, bool SUMS = true>
__global__ __launch_bounds__(push_threads<CTR>()) void push_tiles_kernel(PushArgs<T> a, TileArgs<T> t)
{
    constexpr int kPushThreads = push_threads<CTR>();
    static_assert(FUSE || !SCATTER, "the re-binning launch relies on the census of the fused form");
    constexpr int PPT = Vec16<T>::N;
    constexpr int LW = kPushLds;
    constexpr int SW = kTileLds;
    extern __shared__ __attribute__((aligned(16))) unsigned char push_lds[];
    FPIC_LDS T* lcoef = (FPIC_LDS T*)push_lds;
    FPIC_LDS uint8_t* lsink = (FPIC_LDS uint8_t*)push_lds + static_cast<size_t>(LW) * LW * 12 * sizeof(T);
    FPIC_LDS double* lsums = (FPIC_LDS double*)((FPIC_LDS unsigned char*)push_lds + push_sums_offset<T>());
    FPIC_LDS uint32_t* lcensus = (FPIC_LDS uint32_t*)(lsums + (SUMS ? SW * SW * 4 : 0));
    FPIC_LDS uint32_t* lrank = lcensus + kNbrSlots;
    FPIC_LDS uint32_t* lrange = lrank + kNbrSlots;
    if (blockIdx.x >= *t.nwork) return;
    const BlockWork w = t.work[blockIdx.x];
    const int ti = static_cast<int>(w.tile % t.ntx), tj = static_cast<int>(w.tile / t.ntx);
    const int ti0 = ti * kTileSide, tj0 = tj * kTileSide;
    const int i0 = ti0 - kPushHalo, j0 = tj0 - kPushHalo;
    // stage: one 16-byte piece (a third or a sixth of a record) per lane and iteration
    constexpr int PIECES = static_cast<int>(12 * sizeof(T) / 16);
    using V = typename NatVec16<T>::type;
    for (int k = threadIdx.x; k < LW * LW * PIECES; k += kPushThreads) {
        const int c = k / PIECES, part = k - c * PIECES;
        const int lj = c / LW, li = c - lj * LW;
        const int gi = i0 + li, gj = j0 + lj;
        V val = {};
        if (gi >= 0 && gi < a.nr && gj >= 0 && gj < a.nz)
            val = *reinterpret_cast<const V*>(a.coef + 12 * (static_cast<size_t>(gi) + static_cast<size_t>(a.nr) * gj) + part * Vec16<T>::N);
        *reinterpret_cast<FPIC_LDS V*>(lcoef + 12 * c + part * Vec16<T>::N) = val;
    }
    for (int c = threadIdx.x; c < LW * LW; c += kPushThreads) {
        const int lj = c / LW, li = c - lj * LW;
        const int gi = i0 + li, gj = j0 + lj;
        lsink[c] = (gi >= 0 && gi < a.nr && gj >= 0 && gj < a.nz) ? a.sink_alive[static_cast<size_t>(gi) + static_cast<size_t>(a.nr) * gj] : 0;
    }
    if constexpr (FUSE) {
        if constexpr (SUMS)
            for (int k = threadIdx.x; k < SW * SW * 4; k += kPushThreads) lsums[k] = 0.0;
        if (threadIdx.x < 3 * kNbrSlots) lcensus[threadIdx.x] = 0;
    }
    __syncthreads();
    const WindowTables<T> tab{ GlobalTables<T>{ a.coef, a.sink_alive, a.nr }, lcoef, lsink, i0, j0 };
    const TileNeighbourhood nb{ ti, tj, t.ntx, t.ntz, t.ntiles - 1 };
    unsigned my_spill = 0, my_own = 0;
    const size_t g_begin = (static_cast<size_t>(w.begin) + PPT - 1) / PPT;
    const size_t g_end = (static_cast<size_t>(w.end) + PPT - 1) / PPT;

    if constexpr (!SCATTER) {
        for (size_t g = g_begin + threadIdx.x; g < g_end; g += kPushThreads) {
            const size_t base = g * PPT;
            const int cnt = (base + PPT <= a.n) ? PPT : static_cast<int>(a.n - base);
            Particle<T> q[PPT];
            load_state<T, CTR>(a, base, cnt, q);
            if constexpr (FUSE)
                advance_state<T, CTR>(a, tab, WindowSums<T, SUMS>{ lsums, lcensus, ti0 - kTileHalo, tj0 - kTileHalo, nb, t.cell_sums, t.tile_count, &my_spill, &my_own, a.raster_bits }, cnt, q);
            else
                advance_state<T, CTR>(a, tab, NoSums{}, cnt, q);
            store_state<T, CTR>(a, base, cnt, q);
        }
    } else {
        // Pass A: count the chunk's LOADED states per destination bin (one extra read of
        // x, y, z), then reserve ONE range per bin for the whole chunk.  Everything this
        // workgroup writes into a bin is then one contiguous run (ranks are handed out by
        // LDS atomics in lane order), so cache lines are completed by a single L2 instead of
        // being written piecemeal by many workgroups on several XCDs.
        uint32_t own_count = 0;
        if (t.census_valid) { // counted by the launch before, item for item
            if (threadIdx.x < kNbrSlots) lrank[threadIdx.x] = t.chunk_census[static_cast<size_t>(blockIdx.x) * kNbrSlots + threadIdx.x];
        } else
        for (size_t g = g_begin + threadIdx.x; g < g_end; g += kPushThreads) {
            const size_t base = g * PPT;
            const int cnt = (base + PPT <= a.n) ? PPT : static_cast<int>(a.n - base);
            T px[PPT], py[PPT], pz[PPT];
            load_lane<T, PPT>(a.slab + 0 * a.stride, base, px);
            load_lane<T, PPT>(a.slab + 1 * a.stride, base, py);
            load_lane<T, PPT>(a.slab + 2 * a.stride, base, pz);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (k < cnt) {
                    Particle<T> p;
                    p.x = px[k]; p.y = py[k]; p.z = pz[k];
                    p.r = sqrt_(p.x * p.x + p.y * p.y);
                    int ic = 0, jc = 0;
                    const bool visible = sprite_cell(p, a.nr, a.nz, ic, jc);
                    uint32_t key;
                    const int sl = nb.slot(visible, ic, jc, key);
                    if (sl == kOwnSlot) ++own_count;
                    else if (sl >= 0) __hip_atomic_fetch_add(lrank + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (own_count) __hip_atomic_fetch_add(lrank + kOwnSlot, own_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (threadIdx.x < kNbrSlots) {
            const uint32_t c = lrank[threadIdx.x];
            uint32_t start = 0;
            if (c) {
                const uint32_t bin = nb.bin_of_slot(threadIdx.x);
                start = t.dst_tile_start[bin] + atomicAdd(t.dst_tile_cursor + bin, c);
            }
            lrange[threadIdx.x] = start;
            lrank[threadIdx.x] = 0;
        }
        __syncthreads();
        // Pass B: the push proper; a particle's place is its bin's range + its rank of arrival
        for (size_t g = g_begin + threadIdx.x; g < g_end; g += kPushThreads) {
            const size_t base = g * PPT;
            const int cnt = (base + PPT <= a.n) ? PPT : static_cast<int>(a.n - base);
            Particle<T> q[PPT];
            uint32_t dest[PPT], pid[PPT];
            int slot_of[PPT];
            load_state<T, CTR>(a, base, cnt, q);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                dest[k] = 0; pid[k] = 0; slot_of[k] = -2;
                if (k < cnt) {
                    pid[k] = t.id[base + k];
                    int ic = 0, jc = 0;
                    const bool visible = sprite_cell(q[k], a.nr, a.nz, ic, jc);
                    uint32_t key;
                    const int sl = nb.slot(visible, ic, jc, key);
                    slot_of[k] = sl;
                    if (sl >= 0 && sl != kOwnSlot) dest[k] = lrange[sl] + __hip_atomic_fetch_add(lrank + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else if (sl < 0) dest[k] = t.dst_tile_start[key] + atomicAdd(t.dst_tile_cursor + key, 1u); // beyond the 5x5 tiles: rare
                }
            }
            // stayers: one LDS atomic per wave and k, ranks inside the wave in lane order (consecutive lanes store
            // to consecutive slots)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool own = k < cnt && slot_of[k] == kOwnSlot;
                const unsigned long long mask = __ballot(own);
                if (mask) {
                    const int lane = static_cast<int>(threadIdx.x & 63);
                    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
                    uint32_t wave_base = 0;
                    if (lane == leader) wave_base = __hip_atomic_fetch_add(lrank + kOwnSlot, static_cast<uint32_t>(__popcll(mask)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    wave_base = __shfl(wave_base, leader);
                    if (own) dest[k] = lrange[kOwnSlot] + wave_base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
                }
            }
            advance_state<T, CTR>(a, tab, WindowSums<T, SUMS>{ lsums, lcensus, ti0 - kTileHalo, tj0 - kTileHalo, nb, t.cell_sums, t.tile_count, &my_spill, &my_own, a.raster_bits }, cnt, q);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (k < cnt) {
                    const size_t d = dest[k];
                    t.dst_slab[0 * a.stride + d] = q[k].x; t.dst_slab[1 * a.stride + d] = q[k].y; t.dst_slab[2 * a.stride + d] = q[k].z;
                    t.dst_slab[3 * a.stride + d] = q[k].vx; t.dst_slab[4 * a.stride + d] = q[k].vy; t.dst_slab[5 * a.stride + d] = q[k].vz;
                    if constexpr (!CTR) {
                        t.dst_slab[6 * a.stride + d] = q[k].u1; t.dst_slab[7 * a.stride + d] = q[k].u2;
                        t.dst_slab[8 * a.stride + d] = q[k].c1; t.dst_slab[9 * a.stride + d] = q[k].c2;
                    }
                    t.dst_alive[d] = q[k].alive ? 1 : 0;
                    t.dst_id[d] = pid[k];
                }
            }
        }
    }


    if constexpr (FUSE) {
        if (my_own) __hip_atomic_fetch_add(lcensus + kOwnSlot, my_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        // flush: consecutive lanes take consecutive scalars of one LDS row = consecutive
        // global addresses, so a wave's atomic is one 256-byte piece
#if defined(FPIC_ABL_PUSH) && (FPIC_ABL_PUSH & 1)   // development probe (timing only): the window's sums are not flushed
        if (false)
#endif
        for (int k = threadIdx.x; SUMS && k < SW * SW * 4; k += kPushThreads) {
            const int lj = k / (SW * 4);
            const int rem = k - lj * (SW * 4);
            double val = lsums[k];
            if ((rem & 3) == 3) // integer count -> the sum of that many 0.001's (exact in double)
                val = static_cast<double>(*(FPIC_LDS uint32_t*)(lsums + k)) * static_cast<double>(static_cast<T>(0.001) * static_cast<T>(1));
            if (val == 0.0) continue;
            const int gi = ti0 - kTileHalo + (rem >> 2), gj = tj0 - kTileHalo + lj;
            if (!sums_holds(gi, gj, a.nr, a.nz)) continue;
            atomicAdd(t.cell_sums + 4 * sums_index(gi, gj, a.nr) + (rem & 3), static_cast<T>(val));
        }
        if (threadIdx.x < kNbrSlots) {
            const uint32_t c = lcensus[threadIdx.x];
            const uint32_t bin = nb.bin_of_slot(threadIdx.x);
            if (c && bin != ~0u) atomicAdd(t.tile_count + bin, c);
            if constexpr (!SCATTER) {
                if (t.chunk_census) t.chunk_census[static_cast<size_t>(blockIdx.x) * kNbrSlots + threadIdx.x] = c;
            }
        }
        if (my_spill) atomicAdd(t.spilled, static_cast<unsigned long long>(my_spill));
    }
}

} // namespace fpic
