// fes_kernels.hpp — CDNA4 (gfx950) kernels of the CART3D electrostatic extension:
// the self-consistent push + deposit + field-solve cycle of BASELINE.json configs[2..4].
//
// No reference counterpart (the reference never feeds its deposit back into the fields,
// empic.js:1436-1505): PARITY UNPINNED.  The arithmetic is defined
// by oracle/es3d_oracle_impl.h and followed here operation for operation
// (-ffp-contract=off), so particles, cell indices and the deposited charge agree with
// that oracle bit for bit; the FFT solve agrees within a tolerance.  Conventions kept
// from the reference: positions normalised by the box (empic.js:1199-1244), velocities
// in units of c, h = q dt / 2m (empic.js:44), step factor dt*c (empic.js:852), node
// index i + nx*(j + ny*k) with i fastest (empic.js:1162).
//
// Device layout
//   particles : structure of arrays x,y,z,vx,vy,vz (T) + id (u32), one slab per species
//               and set; 16-byte vector streams, non-temporal; kept binned by 16x16x8-cell tile.
//   E4        : 4 T per node (Ex, Ey, Ez, phi): one 16-byte (float) gather per CIC tap.
//   rho_fixed : int64 per node, 2^42 per unit charge number: the deposit is exact and
//               independent of the order of additions (bit-reproducible, and the sum over
//               ranks of a decomposed run equals the single-GPU grid exactly).
//
// The push workgroup owns one chunk of one tile's particles, stages the tile's field
// records (+2-cell halo) in LDS, gathers with ds_read_b128, pushes, and adds the NEW
// position's eight CIC weights into a second LDS window of int64 accumulators
// (ds_add_u64), flushed once per chunk with 8-byte global atomics: the particle is read
// once and written once per sub-step (48 B fp32 / 96 B fp64 per update).
#pragma once

#include "fes_groups.hpp"
#include "fpic_kernels.hpp"

namespace fes {

using fpic::BlockWork;
using fpic::load_lane;
using fpic::store_lane;
using fpic::NatVec16;
using fpic::Vec16;

#ifndef FPIC_LDS
#define FPIC_LDS __attribute__((address_space(3)))
#endif

// (development switches for the tile shape and the workgroup of the electrostatic push: scripts/probe_push3_tile.sh)
#if !defined(FES_LTX)
#define FES_LTX 4
#endif
#if !defined(FES_LTY)
#define FES_LTY 4
#endif
#if !defined(FES_LTZ)
#define FES_LTZ 3
#endif
#if !defined(FES_PUSH_THREADS)
#define FES_PUSH_THREADS 1024
#endif
constexpr int kTX = 1 << FES_LTX, kTY = 1 << FES_LTY, kTZ = 1 << FES_LTZ;       // cells per tile: 16 x 16 x 8
// The LDS window of a tile: its nodes plus a halo of H cells.  A node costs 16 B of field record + 8 B of
// accumulator in float (H = 2: 21 * 21 * 13 = 5733 nodes, 137.6 KB) and 32 + 8 B in double (H = 1:
// 19 * 19 * 11 = 3971 nodes, 158.8 KB of the CU's 160 KB).
template <typename T>
struct Win {
    static constexpr int H = sizeof(T) == 4 ? 2 : 1;
    static constexpr int X = kTX + 2 * H + 1, Y = kTY + 2 * H + 1, Z = kTZ + 2 * H + 1;
    static constexpr int N = X * Y * Z;
};
constexpr int kPushThreads3 = FES_PUSH_THREADS;                // 16 waves per CU: 14 % faster than 512 (profiles/r02_push3_ablation.txt)
constexpr int kChunk3 = 65536;                   // particles per workgroup and chunk
constexpr int kMaxTiles3 = 40960;                // LDS histogram limit of the one-level binning passes: 160 KB (256^3: 8192 tiles of 16x16x8, 32768 of 8x8x8)
constexpr int kMaxTilesStaged3 = 1 << 20;        // beyond kMaxTiles3 (512^3: 65536 / 262144 tiles): global-atomic census + the staged two-level scatter (<= 1024^2 bins)
constexpr int kFix = 14;                         // fixed-point bits of a CIC weight per axis

// launch shape of a per-node sweep: 256 threads, along x first (a power of two up to FES_NODE_BX), the rest along y; one plane per grid.z
// (64 x 4 nodes per workgroup: a patch shares its y- and z-neighbours' lines; 256 x 1 / 128 x 2 / 64 x 4 / 32 x 8 on one box: the
// gradient 612 / 602 / 577 / 622 us at 512^3, 68 / 62 / 60 / 65 at 256^3 — profiles/r05_node_bx.txt)
#if !defined(FES_NODE_BX)
#define FES_NODE_BX 64u
#endif
struct NodeLaunch {
    dim3 grid, block;
};
inline NodeLaunch node_launch(int nx, int ny, int planes)
{
    unsigned bx = 1;
    while (bx < static_cast<unsigned>(nx) && bx < FES_NODE_BX) bx <<= 1;
    const unsigned by = 256u / bx;
    return { dim3((nx + bx - 1) / bx, (ny + by - 1) / by, static_cast<unsigned>(planes)), dim3(bx, by, 1) };
}


template <typename T> __device__ __forceinline__ T floor_(T v);
template <> __device__ __forceinline__ float floor_<float>(float v) { return floorf(v); }
template <> __device__ __forceinline__ double floor_<double>(double v) { return floor(v); }

// The fused multiply-adds the oracle's definition writes out (es3d_oracle_impl.h: fma(w, E_node, E_p), fma(p, q, -(r*s)),
// fma(dt c / L, v, u)); nothing else is contracted (-ffp-contract=off).
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// two float accumulators at once (v_pk_fma_f32): (acc.x, acc.y) += w * (e.x, e.y), each with one rounding
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2_(float w, f32x2 e, f32x2 acc) { return __builtin_elementwise_fma(f32x2{ w, w }, e, acc); }

template <typename T> __device__ __forceinline__ T fract_(T v);
template <> __device__ __forceinline__ float fract_<float>(float v) { return __builtin_amdgcn_fractf(v); }
template <> __device__ __forceinline__ double fract_<double>(double v) { return __builtin_amdgcn_fract(v); }

// cell and upper weight of a normalised coordinate (es3d_axis).  The same numbers in fewer instructions (the tiled
// kernels are bound by their VALU work: profiles/r03_c3_traffic.json): for 0 <= g < 2^23, g - (T)(int)g is v_fract's
// g - floor(g), exact either way; and "if (i >= n) i -= n" for 0 <= i <= n is the unsigned minimum of i and i - n.
template <typename T>
__device__ __forceinline__ void axis(T u, int n, int& i, int& w1)
{
    const T g = u * static_cast<T>(n);
    i = static_cast<int>(g);
    const T f = fract_(g);
    i = static_cast<int>(min(static_cast<unsigned>(i), static_cast<unsigned>(i - n)));
    w1 = (static_cast<int>(f * static_cast<T>(32768)) + 1) >> 1;
}

// the lower weight as T: (16384 - w1) / 16384 = 1 - w1 / 16384, exact (both are multiples of 2^-14 in [0, 1])
template <typename T>
__device__ __forceinline__ void weights_of(int w1, T (&f)[2])
{
    f[1] = static_cast<T>(w1) * static_cast<T>(1.0 / 16384.0);
    f[0] = static_cast<T>(1) - f[1];
}


template <typename T>
__device__ __forceinline__ T wrap01(T u)
{
    T r = u - floor_(u);
    if (!(r < static_cast<T>(1))) r = static_cast<T>(0);
    return r;
}


template <typename T>
struct Push3Args {
    T* slab;                 // x,y,z,vx,vy,vz, each `stride` elements
    size_t stride;
    unsigned long long n;
    const T* E4;             // node records (Ex,Ey,Ez,phi)
    unsigned long long* rho; // int64 node accumulators (two's complement adds)
    int nx, ny, nz;
    Held held;               // the planes E4 and rho hold
    T hc, tx, ty, tz, sx, sy, sz, dx, dy, dz;
    int Z;                   // charge number of the species
    // tiled form
    int ntx, nty, ntz;
    const BlockWork* work;
    const uint32_t* nwork;
    // a rank of a decomposition pushes the tiles along its slab's faces first (part 1), so that the ghost planes can travel
    // while the interior is pushed (part 2): a work item belongs to the interior when its tile's layer along z
    // (tile / tiles_per_layer) lies in [layer_lo, layer_hi); part 0: the whole list
    int part;
    uint32_t tiles_per_layer, layer_lo, layer_hi;
    unsigned long long* spilled;
    // census of the NEW positions per work item and neighbour slot (27 words per item), written by every in-place launch:
    // the re-binning launch that follows (same work list, same slots) takes its per-bin counts from it instead of
    // counting the chunk again — one read of x, y, z less (nullptr: count).  census_interior_only: a migration since then
    // has changed slots of the layers along the slab's faces (leavers) — their items count, the interior's read
    uint32_t* chunk_census;
    int census_interior_only;
    const uint32_t* tile_start;      // the live bin table: slots [tile_start[t], tile_start[t + 1]) hold tile t
    uint32_t* tile_count;            // census of the NEW positions per tile, zeroed by the host
    // REBIN launch: the other particle set and its bin table
    const uint32_t* id;
    T* dst_slab;
    uint32_t* dst_id;
    const uint32_t* dst_tile_start;
    uint32_t* dst_tile_cursor;
};

template <typename T>
struct P3 {
    T x, y, z, vx, vy, vz;
};

// wx * wy * wzz as int64: the factors are 15-bit weights (wzz: times a charge number of at most 8 bits), so the
// first product is a full-rate 24-bit multiply and the second ONE 32 x 32 -> 64 multiply (64-bit integer
// multiplies are quarter rate on CDNA; written as a chain of them this was a third of the kernel's VALU work)
__device__ __forceinline__ long long weight3(int wx, int wy, int wzz)
{
    return static_cast<long long>(__mul24(wx, wy)) * static_cast<long long>(wzz);
}

// Where a particle's eight field records come from / its eight weights go to.
template <typename T>
struct GlobalGrid {
    const T* E4;
    unsigned long long* rho;
    int nx, ny, nz;
    Held held;
    // LEAN: rolled up, for the tiled kernels' rare particle outside the window (less code and fewer registers there)
    template <bool LEAN = false>
    __device__ __forceinline__ void gather(int i, int j, int k, const T (&fx)[2], const T (&fy)[2], const T (&fz)[2], T& Ex, T& Ey, T& Ez) const
    {
        Ex = Ey = Ez = static_cast<T>(0);
#pragma unroll LEAN ? 1 : 2
        for (int c = 0; c < 2; ++c)
#pragma unroll LEAN ? 1 : 2
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ii = (i + a == nx) ? 0 : i + a, jj = (j + b == ny) ? 0 : j + b, kk = held_plane((k + c == nz) ? 0 : k + c, held, nz);
                    T e[4] = { 0, 0, 0, 0 };
                    if (kk >= 0) fpic::load4(E4 + 4 * (static_cast<size_t>(ii) + static_cast<size_t>(nx) * (static_cast<size_t>(jj) + static_cast<size_t>(ny) * kk)), e);
                    const T w = (fx[a] * fy[b]) * fz[c];
                    Ex = fma_(w, e[0], Ex);
                    Ey = fma_(w, e[1], Ey);
                    Ez = fma_(w, e[2], Ez);
                }
    }
    template <bool LEAN = false>
    __device__ __forceinline__ void deposit(int i, int j, int k, const int (&wx)[2], const int (&wy)[2], const int (&wz)[2], int Z) const
    {
#pragma unroll LEAN ? 1 : 2
        for (int c = 0; c < 2; ++c)
#pragma unroll LEAN ? 1 : 2
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ii = (i + a == nx) ? 0 : i + a, jj = (j + b == ny) ? 0 : j + b, kk = held_plane((k + c == nz) ? 0 : k + c, held, nz);
                    const long long w = weight3(wx[a], wy[b], wz[c] * Z);
                    if (w && kk >= 0) atomicAdd(rho + (static_cast<size_t>(ii) + static_cast<size_t>(nx) * (static_cast<size_t>(jj) + static_cast<size_t>(ny) * kk)),
                                     static_cast<unsigned long long>(w));
                }
    }
};

template <typename T>
__device__ __forceinline__ void load4_lds3(const FPIC_LDS T* p, T (&o)[4])
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

// The tile's window in LDS, global memory behind it for a particle that has left the window.
template <typename T, int ABL = 0>
struct WindowGrid {
    GlobalGrid<T> g;
    const FPIC_LDS T* lE;                  // [Win<T>::N][4]
    FPIC_LDS unsigned long long* lrho;     // [Win<T>::N]
    int ox, oy, oz;                        // node of window slot (0,0,0); may be negative (periodic)
    unsigned* spilled;
    // window slot of cell (i,j,k), or -1 when one of its eight nodes lies outside
    __device__ __forceinline__ int slot(int i, int j, int k) const
    {
        const unsigned l = wrap_near(i - ox, g.nx), m = wrap_near(j - oy, g.ny), n = wrap_near(k - oz, g.nz);
        const bool in = l <= Win<T>::X - 2 && m <= Win<T>::Y - 2 && n <= Win<T>::Z - 2;
        return in ? static_cast<int>(__umul24(__umul24(n, Win<T>::Y) + m, Win<T>::X) + l) : -1; // (24-bit multiplies are full rate)
    }
    __device__ __forceinline__ void gather(int i, int j, int k, const T (&fx)[2], const T (&fy)[2], const T (&fz)[2], T& Ex, T& Ey, T& Ez) const
    {
        const int s = slot(i, j, k);
        if (s < 0) { g.template gather<true>(i, j, k, fx, fy, fz, Ex, Ey, Ez); return; }
        Ex = Ey = Ez = static_cast<T>(0);
        if constexpr ((ABL & 2) != 0) { Ex = fx[0] * fy[1]; Ey = fy[0] * fz[1]; Ez = fz[0] * fx[1]; return; }
        if constexpr (sizeof(T) == 4) {
            // float: the record (Ex, Ey, Ez, phi) of a ds_read_b128 is two register pairs; (Ex, Ey) and (Ez, phi) are
            // accumulated by one v_pk_fma_f32 each (the phi lane is not used)
            f32x2 xy = { 0.f, 0.f }, zp = { 0.f, 0.f };
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        using V = typename NatVec16<T>::type;
                        const V v = *reinterpret_cast<const FPIC_LDS V*>(lE + 4 * (s + a + Win<T>::X * b + Win<T>::X * Win<T>::Y * c));
                        const T w = (fx[a] * fy[b]) * fz[c];
                        xy = fma2_(w, f32x2{ v.x, v.y }, xy);
                        zp = fma2_(w, f32x2{ v.z, v.w }, zp);
                    }
            Ex = xy.x; Ey = xy.y; Ez = zp.x;
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        T e[4];
                        load4_lds3(lE + 4 * (s + a + Win<T>::X * b + Win<T>::X * Win<T>::Y * c), e);
                        const T w = (fx[a] * fy[b]) * fz[c];
                        Ex = fma_(w, e[0], Ex);
                        Ey = fma_(w, e[1], Ey);
                        Ez = fma_(w, e[2], Ez);
                    }
        }
    }
    __device__ __forceinline__ void deposit(int i, int j, int k, const int (&wx)[2], const int (&wy)[2], const int (&wz)[2], int Z) const
    {
        const int s = slot(i, j, k);
        if (s < 0) { g.template deposit<true>(i, j, k, wx, wy, wz, Z); ++*spilled; return; }
        if constexpr ((ABL & 1) != 0) { *spilled += static_cast<unsigned>(wx[0] * wy[1] * wz[0] == 12345); return; }
        const int wzz[2] = { __mul24(wz[0], Z), __mul24(wz[1], Z) };
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const long long w = weight3(wx[a], wy[b], wzz[c]);
                    __hip_atomic_fetch_add(lrho + (s + a + Win<T>::X * b + Win<T>::X * Win<T>::Y * c), static_cast<unsigned long long>(w), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
    }
};

// es3d_push for one particle, then the deposit of its new position (es3d_deposit)
template <typename T, bool HAS_B, typename Grid>
__device__ __forceinline__ void substep3(P3<T>& q, const Push3Args<T>& a, const Grid& grid, int& ni, int& nj, int& nk)
{
    int i, j, k, w1;
    T fx[2], fy[2], fz[2];
    axis(q.x, a.nx, i, w1); weights_of(w1, fx);
    axis(q.y, a.ny, j, w1); weights_of(w1, fy);
    axis(q.z, a.nz, k, w1); weights_of(w1, fz);
    T Ex, Ey, Ez;
    grid.gather(i, j, k, fx, fy, fz, Ex, Ey, Ez);
    const T ax = a.hc * Ex, ay = a.hc * Ey, az = a.hc * Ez;
    T ux = q.vx + ax, uy = q.vy + ay, uz = q.vz + az;
    if constexpr (HAS_B) {
        const T px = ux + fma_(uy, a.tz, -(uz * a.ty));
        const T py = uy + fma_(uz, a.tx, -(ux * a.tz));
        const T pz = uz + fma_(ux, a.ty, -(uy * a.tx));
        const T qx = ux + fma_(py, a.sz, -(pz * a.sy));
        const T qy = uy + fma_(pz, a.sx, -(px * a.sz));
        const T qz = uz + fma_(px, a.sy, -(py * a.sx));
        ux = qx; uy = qy; uz = qz;
    }
    q.vx = ux + ax; q.vy = uy + ay; q.vz = uz + az;
    q.x = wrap01(fma_(a.dx, q.vx, q.x));
    q.y = wrap01(fma_(a.dy, q.vy, q.y));
    q.z = wrap01(fma_(a.dz, q.vz, q.z));
    int wx[2], wy[2], wz[2];
    axis(q.x, a.nx, ni, wx[1]); wx[0] = 16384 - wx[1];
    axis(q.y, a.ny, nj, wy[1]); wy[0] = 16384 - wy[1];
    axis(q.z, a.nz, nk, wz[1]); wz[0] = 16384 - wz[1];
    grid.deposit(ni, nj, nk, wx, wy, wz, a.Z);
}

template <typename T>
__device__ __forceinline__ void load_state3(const Push3Args<T>& a, size_t base, int cnt, P3<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    T v[6][PPT];
#pragma unroll
    for (int f = 0; f < 6; ++f) load_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        q[k].x = v[0][k]; q[k].y = v[1][k]; q[k].z = v[2][k];
        q[k].vx = v[3][k]; q[k].vy = v[4][k]; q[k].vz = v[5][k];
        if (k >= cnt) { q[k].x = q[k].y = q[k].z = static_cast<T>(0.5); q[k].vx = q[k].vy = q[k].vz = static_cast<T>(0); }
    }
}

template <typename T>
__device__ __forceinline__ void store_state3(const Push3Args<T>& a, size_t base, int cnt, const P3<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    T v[6][PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        v[0][k] = q[k].x; v[1][k] = q[k].y; v[2][k] = q[k].z;
        v[3][k] = q[k].vx; v[4][k] = q[k].vy; v[5][k] = q[k].vz;
    }
    if (cnt == PPT) {
#pragma unroll
        for (int f = 0; f < 6; ++f) store_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
    } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (k < cnt) {
#pragma unroll
                for (int f = 0; f < 6; ++f) a.slab[f * a.stride + base + k] = v[f][k];
            }
    }
}

// The same for a group of which the work item owns the slots of `own` only (bit k: slot base + k; fes_groups.hpp,
// groups_exact): a slot of another item is neither used nor written.
template <typename T>
__device__ __forceinline__ unsigned own_mask(uint32_t b0, uint32_t b1, size_t base)
{
    constexpr int PPT = Vec16<T>::N;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) m |= fesgrp::owns(b0, b1, base + k) ? 1u << k : 0u;
    return m;
}
template <typename T>
__device__ __forceinline__ void load_state3_own(const Push3Args<T>& a, size_t base, unsigned own, P3<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    T v[6][PPT];
#pragma unroll
    for (int f = 0; f < 6; ++f) load_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        q[k].x = v[0][k]; q[k].y = v[1][k]; q[k].z = v[2][k];
        q[k].vx = v[3][k]; q[k].vy = v[4][k]; q[k].vz = v[5][k];
        if (!((own >> k) & 1u)) { q[k].x = q[k].y = q[k].z = static_cast<T>(0.5); q[k].vx = q[k].vy = q[k].vz = static_cast<T>(0); }
    }
}
template <typename T>
__device__ __forceinline__ void store_state3_own(const Push3Args<T>& a, size_t base, unsigned own, const P3<T> (&q)[Vec16<T>::N])
{
    constexpr int PPT = Vec16<T>::N;
    T v[6][PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        v[0][k] = q[k].x; v[1][k] = q[k].y; v[2][k] = q[k].z;
        v[3][k] = q[k].vx; v[4][k] = q[k].vy; v[5][k] = q[k].vz;
    }
    if (own == (1u << PPT) - 1u) {
#pragma unroll
        for (int f = 0; f < 6; ++f) store_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
    } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if ((own >> k) & 1u) {
#pragma unroll
                for (int f = 0; f < 6; ++f) a.slab[f * a.stride + base + k] = v[f][k];
            }
    }
}

static_assert((kTX & (kTX - 1)) == 0 && (kTY & (kTY - 1)) == 0 && (kTZ & (kTZ - 1)) == 0, "tile edges are powers of two: cell -> tile is a shift");
// LX, LY, LZ: log2 of the tile edges (the electrostatic cycle bins by 16x16x8 cells, the full-EM cycle by 8x8x8)
template <int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
__device__ __forceinline__ uint32_t tile_key3(int i, int j, int k, int ntx, int nty)
{
    const unsigned tx = static_cast<unsigned>(i) >> LX, ty = static_cast<unsigned>(j) >> LY, tz = static_cast<unsigned>(k) >> LZ;
    return tx + __umul24(static_cast<unsigned>(ntx), ty + __umul24(static_cast<unsigned>(nty), tz));
}

// the tile of a position
template <typename T, int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
__device__ __forceinline__ uint32_t key_of(T x, T y, T z, int nx, int ny, int nz, int ntx, int nty)
{
    int i, j, k, w;
    axis(x, nx, i, w); axis(y, ny, j, w); axis(z, nz, k, w);
    return tile_key3<LX, LY, LZ>(i, j, k, ntx, nty);
}

// Flat form: any particle order, everything through global memory (L2 gathers, 8-byte global
// atomics).  Used until the particles have been binned.  DEPOSIT_ONLY: no push, the deposit of
// the CURRENT positions (precalc()).
template <typename T, bool HAS_B, bool DEPOSIT_ONLY>
__global__ __launch_bounds__(256) void push3_flat_kernel(Push3Args<T> a)
{
    constexpr int PPT = Vec16<T>::N;
    const size_t base = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * PPT;
    if (base >= a.n) return;
    const int cnt = (base + PPT <= a.n) ? PPT : static_cast<int>(a.n - base);
    const GlobalGrid<T> grid{ a.E4, a.rho, a.nx, a.ny, a.nz, a.held };
    P3<T> q[PPT];
    load_state3(a, base, cnt, q);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        if (k >= cnt) continue;
        int ni, nj, nk;
        if constexpr (DEPOSIT_ONLY) {
            int wx[2], wy[2], wz[2];
            axis(q[k].x, a.nx, ni, wx[1]); wx[0] = 16384 - wx[1];
            axis(q[k].y, a.ny, nj, wy[1]); wy[0] = 16384 - wy[1];
            axis(q[k].z, a.nz, nk, wz[1]); wz[0] = 16384 - wz[1];
            grid.deposit(ni, nj, nk, wx, wy, wz, a.Z);
        } else {
            substep3<T, HAS_B>(q[k], a, grid, ni, nj, nk);
        }
    }
    if constexpr (!DEPOSIT_ONLY) store_state3(a, base, cnt, q);
}

// A workgroup's bumps of per-tile counters, gathered in a 32-entry LDS table first: the particles a migration moves sit
// in a few tiles per workgroup and hundreds per tile, and as many global atomics on one counter serialise.  tally_add:
// any thread, any time between tally_reset and tally_flush (both called by every thread of the workgroup).
constexpr int kTallySlots = 32;
struct TileTally {
    uint32_t key[kTallySlots], val[kTallySlots];
};
__device__ __forceinline__ void tally_reset(TileTally& t)
{
    if (threadIdx.x < kTallySlots) { t.key[threadIdx.x] = ~0u; t.val[threadIdx.x] = 0; }
    __syncthreads();
}
template <bool ADD>
__device__ __forceinline__ void tally_add(TileTally& t, uint32_t key, uint32_t* __restrict__ counters)
{
    uint32_t h = key & (kTallySlots - 1);
    for (int probe = 0; probe < kTallySlots; ++probe, h = (h + 1) & (kTallySlots - 1)) {
        const uint32_t old = atomicCAS(&t.key[h], ~0u, key);
        if (old == ~0u || old == key) { atomicAdd(&t.val[h], 1u); return; }
    }
    if (ADD) atomicAdd(counters + key, 1u); else atomicSub(counters + key, 1u); // (table full: straight to the counter)
}
template <bool ADD>
__device__ __forceinline__ void tally_flush(TileTally& t, uint32_t* __restrict__ counters)
{
    __syncthreads();
    if (threadIdx.x < kTallySlots && t.key[threadIdx.x] != ~0u && t.val[threadIdx.x]) {
        if (ADD) atomicAdd(counters + t.key[threadIdx.x], t.val[threadIdx.x]); else atomicSub(counters + t.key[threadIdx.x], t.val[threadIdx.x]);
    }
}

// Runs of equal keys over consecutive lanes of a wavefront (every lane of the wavefront calls this): the run's first
// lane and its length.  A migration's records arrive almost in tile order, so that the per-tile counters they bump see one
// atomic per run instead of one per record (hundreds of records per counter: contended atomics were most of those passes).
__device__ __forceinline__ void wave_runs(uint32_t key, int& head_lane, int& length)
{
    const int lane = static_cast<int>(threadIdx.x & 63);
    const uint32_t prev = __shfl_up(key, 1);
    const unsigned long long heads = __ballot(lane == 0 || key != prev);
    head_lane = 63 - __clzll(static_cast<long long>(heads & (~0ull >> (63 - lane))));
    const unsigned long long rest = head_lane == 63 ? 0ull : heads >> (head_lane + 1);
    length = rest ? 1 + (__ffsll(static_cast<long long>(rest)) - 1) : 64 - head_lane;
}

// The arrivals of a migration (appended behind the sorted array, in no order) inside a re-binning launch: pushed
// against the global grid, stored in the other particle set at their bin (the tile of the position they arrived
// with, which the migration added to the census), counted in the census of the new positions.
template <typename T, bool HAS_B>
__global__ __launch_bounds__(256) void push3_tail_kernel(Push3Args<T> a, size_t first, size_t count)
{
    const size_t r = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = r < count;
    const size_t s = first + (active ? r : 0);
    const int lane = static_cast<int>(threadIdx.x & 63);
    const GlobalGrid<T> grid{ a.E4, a.rho, a.nx, a.ny, a.nz, a.held };
    P3<T> q;
    q.x = a.slab[s]; q.y = a.slab[a.stride + s]; q.z = a.slab[2 * a.stride + s];
    q.vx = a.slab[3 * a.stride + s]; q.vy = a.slab[4 * a.stride + s]; q.vz = a.slab[5 * a.stride + s];
    // the arrivals come almost in tile order: a run of one tile's records takes its places with ONE returning atomic
    const uint32_t bin = active ? key_of<T>(q.x, q.y, q.z, a.nx, a.ny, a.nz, a.ntx, a.nty) : ~0u;
    int head, len;
    wave_runs(bin, head, len);
    uint32_t run_base = 0;
    if (active && head == lane) run_base = atomicAdd(a.dst_tile_cursor + bin, static_cast<uint32_t>(len));
    run_base = __shfl(run_base, head);
    __shared__ TileTally tally;
    tally_reset(tally);
    if (active) {
        const size_t d = static_cast<size_t>(a.dst_tile_start[bin]) + run_base + static_cast<uint32_t>(lane - head);
        int ni, nj, nk;
        substep3<T, HAS_B>(q, a, grid, ni, nj, nk);
        tally_add<true>(tally, tile_key3<4, 4, 3>(ni, nj, nk, a.ntx, a.nty), a.tile_count); // (the census of the new positions)
        a.dst_slab[d] = q.x; a.dst_slab[a.stride + d] = q.y; a.dst_slab[2 * a.stride + d] = q.z;
        a.dst_slab[3 * a.stride + d] = q.vx; a.dst_slab[4 * a.stride + d] = q.vy; a.dst_slab[5 * a.stride + d] = q.vz;
        a.dst_id[d] = a.id[s];
    }
    tally_flush<true>(tally, a.tile_count);
}

// A workgroup tracks its own tile and the 26 around it (periodic) in LDS when it counts or ranks particles
// by tile; a particle further away goes to the global tables directly.
constexpr int kNbr3 = 27;
struct Neighbourhood3 {
    int ti, tj, tk, ntx, nty, ntz;
    // per axis: 0 = own tile, 1 = next, 2 = previous (periodic); -1 = further away
    static __device__ __forceinline__ int rel(int t, int own, int nt)
    {
        int d = t - own;
        if (d < 0) d += nt;
        return d == 0 ? 0 : (d == 1 ? 1 : (d == nt - 1 ? 2 : -1));
    }
    __device__ __forceinline__ int slot(int i, int j, int k, uint32_t& key) const
    {
        const int tx = static_cast<int>(static_cast<unsigned>(i) / kTX), ty = static_cast<int>(static_cast<unsigned>(j) / kTY),
                  tz = static_cast<int>(static_cast<unsigned>(k) / kTZ);
        key = static_cast<uint32_t>(tx) + __umul24(static_cast<unsigned>(ntx), static_cast<unsigned>(ty) + __umul24(static_cast<unsigned>(nty), static_cast<unsigned>(tz)));
        const int rx = rel(tx, ti, ntx), ry = rel(ty, tj, nty), rz = rel(tz, tk, ntz);
        return (rx | ry | rz) < 0 ? -1 : rx + 3 * (ry + 3 * rz);
    }
    // global bin of an LDS slot, or ~0u when two slots would name the same tile (fewer than 3 tiles along an axis)
    __device__ __forceinline__ uint32_t bin_of_slot(int s) const
    {
        const int rx = s % 3, ry = (s / 3) % 3, rz = s / 9;
        if ((rx == 2 && ntx < 3) || (ry == 2 && nty < 3) || (rz == 2 && ntz < 3)) return ~0u;
        if ((rx == 1 && ntx < 2) || (ry == 1 && nty < 2) || (rz == 1 && ntz < 2)) return ~0u;
        const int tx = (ti + (rx == 2 ? ntx - 1 : rx)) % ntx, ty = (tj + (ry == 2 ? nty - 1 : ry)) % nty, tz = (tk + (rz == 2 ? ntz - 1 : rz)) % ntz;
        return static_cast<uint32_t>(tx) + static_cast<uint32_t>(ntx) * (static_cast<uint32_t>(ty) + static_cast<uint32_t>(nty) * static_cast<uint32_t>(tz));
    }
};

// does the work item of this tile belong to the part being launched?  (0: every item; 2: the slab's interior tile layers;
// 1: the layers along the faces and the ghost layers beyond them.)  Decided from the item itself and launch constants:
// every workgroup of either launch sees the same partition whatever else is in flight.
__device__ __forceinline__ bool in_part(uint32_t tile, int part, uint32_t tiles_per_layer, uint32_t layer_lo, uint32_t layer_hi)
{
    if (part == 0) return true;
    const uint32_t layer = tile / tiles_per_layer;
    const bool interior = layer >= layer_lo && layer < layer_hi;
    return (part == 2) == interior;
}


// One launch pushes every species of a handle: a workgroup stages its tile's window once, takes the tile's particles of
// each species in turn — all of them deposit into the one accumulator window — and flushes once.  At 512^3 / 2e9 (15 000
// particles per tile and species) the windows of two launches were 12 + 6 GB per sub-step beside 96 GB of particles.
// A joint work item is (tile, k): the k-th piece of `chunk` slots of the tile in every species' bin table; chunk == 0: one
// species with its own list of items (tile, begin, end).
constexpr int kJointMax = 4;
template <typename T>
struct Push3Joint {
    Push3Args<T> sp[kJointMax];   // (the grid, the decomposition and the launch constants are the same in all of them)
    int nsp;
    const BlockWork* work;
    const uint32_t* nwork;
    uint32_t chunk;
};

// the joint work list from the species' bin tables: tile t contributes max over the species of ceil(n_s(t) / chunk) items
// (tile, k), in tile order.  One workgroup of 1024 threads (the tables have at most 2^20 tiles).
struct JointTables {
    const uint32_t* tile_start[kJointMax];
    int n;
};
__global__ __launch_bounds__(1024) void joint_scan_kernel(JointTables tabs, uint32_t ntiles, uint32_t chunk, BlockWork* __restrict__ work, uint32_t* __restrict__ nwork)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (ntiles + 1023u) / 1024u;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    auto items_of = [&](uint32_t t) {
        uint32_t m = 0;
        for (int s = 0; s < tabs.n; ++s) {
            const uint32_t k = fesgrp::pieces_of(tabs.tile_start[s][t + 1] - tabs.tile_start[s][t], chunk);
            m = k > m ? k : m;
        }
        return m;
    };
    uint32_t mine = 0;
    for (uint32_t t = t0; t < t1; ++t) mine += items_of(t);
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // inclusive scan
        const uint32_t v = threadIdx.x >= static_cast<unsigned>(d) ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t at = part[threadIdx.x] - mine;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t m = items_of(t);
        for (uint32_t k = 0; k < m; ++k) work[at++] = BlockWork{ t, k, 0u, 0u };
    }
    if (threadIdx.x == 1023) *nwork = part[1023];
}

// The slots [b0, b1) of a work item in one species and the groups of PPT slots that hold them: the item pushes exactly its
// own slots (fes_groups.hpp, groups_exact — since round 4; before, a group that straddled two items went to the earlier one,
// whose workgroup then pushed up to PPT - 1 particles of ANOTHER tile through global memory at the end of every tile).
template <typename T>
__device__ __forceinline__ void species_groups(const Push3Joint<T>& j, const Push3Args<T>& a, const BlockWork& w, int ppt, uint32_t& b0, uint32_t& b1, size_t& g_begin,
                                               size_t& g_end)
{
    b0 = w.begin; b1 = w.end;
    if (j.chunk) fesgrp::piece(a.tile_start[w.tile], a.tile_start[w.tile + 1], w.begin, j.chunk, b0, b1);
    fesgrp::groups_exact(b0, b1, ppt, g_begin, g_end);
}


template <typename T>
constexpr size_t push3_lds_bytes() { return static_cast<size_t>(Win<T>::N) * (4 * sizeof(T) + 8) + 3 * kNbr3 * sizeof(uint32_t) + 16; }

// Tiled form for binned particles: one workgroup per work item — a piece of one tile's particles, of every species of
// the launch in turn (Push3Joint): the window is staged once, every species adds into it, it is flushed once.
//
// Every launch also counts the NEW positions per tile (census): the table the next re-binning is
// laid out from.  REBIN makes this launch the re-binning as well: the bin of a particle is the tile
// of the position it is LOADED with — exactly what the previous launch's census counted, from which
// the host has laid out dst_tile_start.  The workgroup first counts its chunk per destination bin
// (one extra read of x, y, z), reserves ONE range per bin, then pushes and stores each particle at
// range + rank of arrival in the other particle set: no separate count / scatter passes in steady state.
//
// ABL: development probe bits, timing only (scripts/ablate_push3.hip): 1 = no LDS accumulation, 2 = no field
// gather, 4 = no window staging / flush, 8 = the flush stores instead of adding atomically.  The library instantiates ABL = 0 only.
template <typename T, bool HAS_B, bool DEPOSIT_ONLY, bool REBIN = false, int THREADS = kPushThreads3, int ABL = 0>
__global__ __launch_bounds__(THREADS) void push3_tiles_kernel(Push3Joint<T> J)
{
    static_assert(!(REBIN && DEPOSIT_ONLY), "precalc() never re-bins");
    constexpr int PPT = Vec16<T>::N;
    constexpr int kWX = Win<T>::X, kWY = Win<T>::Y, kWN = Win<T>::N, kHalo = Win<T>::H;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    FPIC_LDS T* lE = (FPIC_LDS T*)lds3;
    FPIC_LDS unsigned long long* lrho = (FPIC_LDS unsigned long long*)((FPIC_LDS unsigned char*)lds3 + static_cast<size_t>(kWN) * 4 * sizeof(T));
    FPIC_LDS uint32_t* lcensus = (FPIC_LDS uint32_t*)(lrho + kWN);
    FPIC_LDS uint32_t* lrank = lcensus + kNbr3;
    FPIC_LDS uint32_t* lrange = lrank + kNbr3;
    const Push3Args<T>& c0 = J.sp[0]; // (what is the same for every species is read from the first)
    // (the item is read before the count that may send the workgroup home: one memory latency instead of two; the list
    // has a slot for every workgroup of the launch)
    const BlockWork w = J.work[blockIdx.x];
    if (blockIdx.x >= *J.nwork) return;
    const uint32_t wi = blockIdx.x;
    if (!in_part(w.tile, c0.part, c0.tiles_per_layer, c0.layer_lo, c0.layer_hi)) return;
    const int ti = static_cast<int>(w.tile % c0.ntx), tj = static_cast<int>((w.tile / c0.ntx) % c0.nty), tk = static_cast<int>(w.tile / (c0.ntx * c0.nty));
    const int ox = ti * kTX - kHalo, oy = tj * kTY - kHalo, oz = tk * kTZ - kHalo;
    const Neighbourhood3 nb{ ti, tj, tk, c0.ntx, c0.nty, c0.ntz };
    // stage the window: one node record (16 B float / 32 B double) per lane and iteration
    using V = typename NatVec16<T>::type;
    constexpr int PIECES = static_cast<int>(4 * sizeof(T) / 16);
    for (int s = threadIdx.x; s < ((ABL & 4) ? 0 : kWN); s += THREADS) {
        const int n = s / (kWX * kWY), rem = s - n * (kWX * kWY);
        const int m = rem / kWX, l = rem - m * kWX;
        // (wrap_window: no division by a run-time number for boxes of 32 nodes or more)
        const int gi = wrap_window(ox + l, c0.nx), gj = wrap_window(oy + m, c0.ny),
                  gk = wrap_window(oz - c0.held.zs0 + n, c0.nz); // gk: among the planes held
        lrho[s] = 0ull;
        if constexpr (!DEPOSIT_ONLY) {
            const int lk = gk < c0.held.nzs ? gk : -1;
            if (lk >= 0) {
                const T* src = c0.E4 + 4 * (static_cast<size_t>(gi) + static_cast<size_t>(c0.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(c0.ny) * lk));
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    *reinterpret_cast<FPIC_LDS V*>(lE + 4 * s + p * Vec16<T>::N) = *reinterpret_cast<const V*>(src + p * Vec16<T>::N);
            } else { // a plane the rank does not hold (beyond its ghost planes): no particle of the tile gathers there
#pragma unroll
                for (int c = 0; c < 4; ++c) lE[4 * s + c] = static_cast<T>(0);
            }
        }
    }
    if (threadIdx.x < 3 * kNbr3) lcensus[threadIdx.x] = 0;
    __syncthreads();
    unsigned my_spill = 0;
    const WindowGrid<T, ABL> grid{ GlobalGrid<T>{ c0.E4, c0.rho, c0.nx, c0.ny, c0.nz, c0.held }, lE, lrho, ox, oy, oz, &my_spill };
    for (int sp = 0; sp < J.nsp; ++sp) {
    const Push3Args<T>& a = J.sp[sp];
    uint32_t census_own = 0;
    size_t g_begin, g_end;
    uint32_t b0, b1;
    species_groups(J, a, w, PPT, b0, b1, g_begin, g_end);

    if constexpr (REBIN) {
        // Pass A: the chunk's LOADED positions per destination bin — what the launch before counted as its new positions
        // (chunk_census), or counted now — then one range per bin is reserved
        uint32_t own_count = 0;
        bool counted_before = a.chunk_census != nullptr;
        if (counted_before && a.census_interior_only) {
            const uint32_t layer = w.tile / a.tiles_per_layer;
            counted_before = layer >= a.layer_lo && layer < a.layer_hi;
        }
        if (counted_before) {
            if (threadIdx.x < kNbr3) lrank[threadIdx.x] = a.chunk_census[static_cast<size_t>(wi) * kNbr3 + threadIdx.x];
        } else
        for (size_t g = g_begin + threadIdx.x; g < g_end; g += THREADS) {
            const size_t base = g * PPT;
            const unsigned own = own_mask<T>(b0, b1, base);
            T px[PPT], py[PPT], pz[PPT];
            load_lane<T, PPT>(a.slab + 0 * a.stride, base, px);
            load_lane<T, PPT>(a.slab + 1 * a.stride, base, py);
            load_lane<T, PPT>(a.slab + 2 * a.stride, base, pz);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (!((own >> k) & 1u) || px[k] < static_cast<T>(0)) continue; // (x < 0: the slot of a particle that has migrated away)
                int i, j, kk, wgt;
                axis(px[k], a.nx, i, wgt); axis(py[k], a.ny, j, wgt); axis(pz[k], a.nz, kk, wgt);
                uint32_t key;
                const int sl = nb.slot(i, j, kk, key);
                // nearly every particle is still in its own tile: those are counted in a register (64 lanes adding
                // to ONE LDS word would be serialised), the few leavers by LDS atomics
                if (sl == 0) ++own_count;
                else if (sl > 0) __hip_atomic_fetch_add(lrank + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (own_count) __hip_atomic_fetch_add(lrank, own_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        if (threadIdx.x < kNbr3) {
            const uint32_t c = lrank[threadIdx.x];
            uint32_t start = 0;
            if (c) {
                const uint32_t bin = nb.bin_of_slot(threadIdx.x);
                start = a.dst_tile_start[bin] + atomicAdd(a.dst_tile_cursor + bin, c);
            }
            lrange[threadIdx.x] = start;
            lrank[threadIdx.x] = 0;
        }
        __syncthreads();
    }

    for (size_t g = g_begin + threadIdx.x; g < g_end; g += THREADS) {
        const size_t base = g * PPT;
        const unsigned own = own_mask<T>(b0, b1, base);
        P3<T> q[PPT];
        uint32_t dest[PPT], pid[PPT];
        int slot_of[PPT];
        load_state3_own(a, base, own, q);
        if constexpr (REBIN) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                dest[k] = 0; pid[k] = 0; slot_of[k] = -2;
                if (!((own >> k) & 1u) || q[k].x < static_cast<T>(0)) continue;
                pid[k] = a.id[base + k];
                int i, j, kk, wgt;
                axis(q[k].x, a.nx, i, wgt); axis(q[k].y, a.ny, j, wgt); axis(q[k].z, a.nz, kk, wgt);
                uint32_t key;
                const int sl = nb.slot(i, j, kk, key);
                slot_of[k] = sl;
                if (sl > 0) dest[k] = lrange[sl] + __hip_atomic_fetch_add(lrank + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (sl < 0) dest[k] = a.dst_tile_start[key] + atomicAdd(a.dst_tile_cursor + key, 1u); // beyond the 27 tiles: rare
            }
            // stayers: one LDS atomic per wave and k, ranks inside the wave by lane order (consecutive lanes ->
            // consecutive destinations: the stores below coalesce)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool stays = slot_of[k] == 0; // (-2 for a slot that is not the item's, or dead)
                const unsigned long long mask = __ballot(stays);
                if (mask) {
                    const int lane = static_cast<int>(threadIdx.x & 63);
                    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
                    uint32_t wave_base = 0;
                    if (lane == leader) wave_base = __hip_atomic_fetch_add(lrank, static_cast<uint32_t>(__popcll(mask)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    wave_base = __shfl(wave_base, leader);
                    if (stays) dest[k] = lrange[0] + wave_base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((own >> k) & 1u)) continue;
            if constexpr (REBIN) { if (slot_of[k] == -2) continue; } // a dead slot
            int ni, nj, nk;
            if constexpr (DEPOSIT_ONLY) {
                int wx[2], wy[2], wz[2];
                axis(q[k].x, a.nx, ni, wx[1]); wx[0] = 16384 - wx[1];
                axis(q[k].y, a.ny, nj, wy[1]); wy[0] = 16384 - wy[1];
                axis(q[k].z, a.nz, nk, wz[1]); wz[0] = 16384 - wz[1];
                grid.deposit(ni, nj, nk, wx, wy, wz, a.Z);
            } else {
                substep3<T, HAS_B>(q[k], a, grid, ni, nj, nk);
                // census of the new positions per tile
                uint32_t key;
                const int sl = nb.slot(ni, nj, nk, key);
                if (sl == 0) ++census_own;      // (own tile: a register, see pass A)
                else if (sl > 0) __hip_atomic_fetch_add(lcensus + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else atomicAdd(a.tile_count + key, 1u);
            }
        }
        if constexpr (REBIN) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (slot_of[k] == -2) continue; // (not the item's, or dead)
                const size_t d = dest[k];
                a.dst_slab[0 * a.stride + d] = q[k].x; a.dst_slab[1 * a.stride + d] = q[k].y; a.dst_slab[2 * a.stride + d] = q[k].z;
                a.dst_slab[3 * a.stride + d] = q[k].vx; a.dst_slab[4 * a.stride + d] = q[k].vy; a.dst_slab[5 * a.stride + d] = q[k].vz;
                a.dst_id[d] = pid[k];
            }
        } else if constexpr (!DEPOSIT_ONLY) {
            store_state3_own(a, base, own, q);
        }
    }
    if (census_own) __hip_atomic_fetch_add(lcensus, census_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if constexpr (!DEPOSIT_ONLY) {
        if (threadIdx.x < kNbr3) {
            const uint32_t c = lcensus[threadIdx.x];
            const uint32_t bin = nb.bin_of_slot(threadIdx.x);
            if (c && bin != ~0u) atomicAdd(a.tile_count + bin, c);
            if constexpr (!REBIN) {
                if (a.chunk_census) a.chunk_census[static_cast<size_t>(wi) * kNbr3 + threadIdx.x] = c;
            }
        }
    }
    if (sp + 1 < J.nsp) { // the next species counts from zero
        __syncthreads();
        if (threadIdx.x < 3 * kNbr3) lcensus[threadIdx.x] = 0;
        __syncthreads();
    }
    } // species
    // flush the non-zero accumulators: consecutive lanes take consecutive slots of one window row
    for (int s = threadIdx.x; s < ((ABL & 4) ? 0 : kWN); s += THREADS) {
        const unsigned long long v = lrho[s];
        if (v == 0ull) continue;
        const int n = s / (kWX * kWY), rem = s - n * (kWX * kWY);
        const int m = rem / kWX, l = rem - m * kWX;
        // (wrap_window: no division by a run-time number for boxes of 32 nodes or more)
        const int gi = wrap_window(ox + l, c0.nx), gj = wrap_window(oy + m, c0.ny),
                  gk = wrap_window(oz - c0.held.zs0 + n, c0.nz);
        if constexpr ((ABL & 8) != 0) { // development probe (timing only): plain stores where the library adds atomically
            if (gk < c0.held.nzs) c0.rho[static_cast<size_t>(gi) + static_cast<size_t>(c0.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(c0.ny) * gk)] = v;
            continue;
        }
        if (gk < c0.held.nzs) atomicAdd(c0.rho + (static_cast<size_t>(gi) + static_cast<size_t>(c0.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(c0.ny) * gk)), v);
    }
    if (my_spill) atomicAdd(c0.spilled, static_cast<unsigned long long>(my_spill));
}

// ------------------------------------------------------------------ field solve

// rho[node] = (T)((double)fixed * scale)   (es3d_rho_real)
template <typename T>
__global__ __launch_bounds__(256) void rho_real_kernel(const long long* __restrict__ fixed, size_t nodes, double scale, T* __restrict__ rho)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < nodes) rho[c] = static_cast<T>(static_cast<double>(fixed[c]) * scale);
}

// phi_hat = rho_hat / (eps0 K^2 N) on the half spectrum of the real transform ((nx/2+1) x ny x nz,
// interleaved complex); k2x/k2y/k2z are the per-axis eigenvalue tables in double; the mean mode is 0
template <typename T>
__global__ __launch_bounds__(256) void kspace_kernel(T* __restrict__ hat, int nxh, int ny, int nz, const double* __restrict__ k2x,
                                                     const double* __restrict__ k2y, const double* __restrict__ k2z, double inv_eps0_n)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nxh) * ny * nz) return;
    const int i = static_cast<int>(c % nxh);
    const int j = static_cast<int>((c / nxh) % ny);
    const int k = static_cast<int>(c / (static_cast<size_t>(nxh) * ny));
    const double K2 = (k2x[i] + k2y[j]) + k2z[k];
    const T g = (i | j | k) ? static_cast<T>(inv_eps0_n / K2) : static_cast<T>(0);
    hat[2 * c] = hat[2 * c] * g;
    hat[2 * c + 1] = hat[2 * c + 1] * g;
}

// node records (Ex,Ey,Ez,phi) from phi by central differences, periodic (es3d_gradient)
template <typename T>
__global__ __launch_bounds__(256) void gradient_kernel(const T* __restrict__ phi, int nx, int ny, int nz, T hx, T hy, T hz, T* __restrict__ E4)
{
    // (node_launch: threads along x, then y, the plane on grid.z — a flat index cost three 64-bit divisions by run-time numbers
    // per node, more instructions than the sweep's seven loads and one store: 661 -> 5xx us at 512^3, profiles/r05_fft_vec.txt)
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x), j = static_cast<int>(blockIdx.y * blockDim.y + threadIdx.y), k = static_cast<int>(blockIdx.z);
    if (i >= nx || j >= ny) return;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    const size_t c = i + sy * j + sz * k;
    const int im = i ? i - 1 : nx - 1, ip = (i + 1 == nx) ? 0 : i + 1;
    const int jm = j ? j - 1 : ny - 1, jp = (j + 1 == ny) ? 0 : j + 1;
    const int km = k ? k - 1 : nz - 1, kp = (k + 1 == nz) ? 0 : k + 1;
    using V = typename NatVec16<T>::type;
    T o[4];
    o[0] = (phi[im + sy * j + sz * k] - phi[ip + sy * j + sz * k]) * hx;
    o[1] = (phi[i + sy * jm + sz * k] - phi[i + sy * jp + sz * k]) * hy;
    o[2] = (phi[i + sy * j + sz * km] - phi[i + sy * j + sz * kp]) * hz;
    o[3] = phi[c];
    if constexpr (sizeof(T) == 4) {
        V v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(E4 + 4 * c)); // (read next by the push's staging, 2 GB later at 512^3)
    } else {
        V v0, v1; v0.x = o[0]; v0.y = o[1]; v1.x = o[2]; v1.y = o[3];
        __builtin_nontemporal_store(v0, reinterpret_cast<V*>(E4 + 4 * c));
        __builtin_nontemporal_store(v1, reinterpret_cast<V*>(E4 + 4 * c + 2));
    }
}

// ------------------------------------------------------------------ uploads, read-back, binning

// out.set({position}) for the box: u = (T)(x * (1/L)) wrapped into [0,1) (es3d_normalise)
template <typename T, typename In>
__global__ __launch_bounds__(256) void set_pos3_kernel(const In* __restrict__ aos, size_t chunk_begin, size_t chunk_n, double fx, double fy, double fz,
                                                       T* x, T* y, T* z, const uint32_t* __restrict__ id, size_t n, size_t slot0 = 0)
{
    // (id == nullptr: identity order, only the chunk's own slots [slot0, n) are visited; see set_vec3_kernel)
    const size_t s = slot0 + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = id ? id[s] : s;
    if (i < chunk_begin || i >= chunk_begin + chunk_n) return;
    const In* v = aos + 3 * (i - chunk_begin);
    x[s] = wrap01(static_cast<T>(static_cast<double>(v[0]) * fx));
    y[s] = wrap01(static_cast<T>(static_cast<double>(v[1]) * fy));
    z[s] = wrap01(static_cast<T>(static_cast<double>(v[2]) * fz));
}

// out.set({E}) for the box: value[i][j][k][c] -> node record i + nx*(j + ny*k), phi = 0
template <typename T, typename In>
__global__ __launch_bounds__(256) void pack_field3_kernel(const In* __restrict__ in, int nx, int ny, int nz, T* __restrict__ E4, Held held)
{
    const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= static_cast<size_t>(nx) * ny * nz) return;
    const int i = static_cast<int>(g % nx), j = static_cast<int>((g / nx) % ny), k = static_cast<int>(g / (static_cast<size_t>(nx) * ny));
    const int lk = held_plane(k, held, nz);
    if (lk < 0) return; // (a compact rank keeps the planes it holds)
    const size_t c = i + static_cast<size_t>(nx) * (j + static_cast<size_t>(ny) * lk);
    const In* v = in + 3 * ((static_cast<size_t>(i) * ny + j) * nz + k);
    E4[4 * c] = static_cast<T>(v[0]); E4[4 * c + 1] = static_cast<T>(v[1]); E4[4 * c + 2] = static_cast<T>(v[2]); E4[4 * c + 3] = static_cast<T>(0);
}

template <typename T>
__global__ __launch_bounds__(256) void cells3_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                                                     const uint32_t* __restrict__ id, size_t n, size_t chunk_begin, size_t chunk_n, int nx, int ny,
                                                     int nz, int32_t* __restrict__ cells, size_t first = 0, size_t stride = 1)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    size_t p = id[s];
    if (p < first || (p - first) % stride) return;   // slot k = the caller's particle first + k * stride (get_vec3_kernel)
    p = (p - first) / stride;
    if (p < chunk_begin || p >= chunk_begin + chunk_n) return;
    int i, j, k, w;
    axis(x[s], nx, i, w); axis(y[s], ny, j, w); axis(z[s], nz, k, w);
    cells[p - chunk_begin] = i + nx * (j + ny * k);
}

template <typename T>
__global__ __launch_bounds__(256) void init3_kernel(T* slab, size_t stride, uint32_t* id)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= stride) return;
#pragma unroll
    for (int f = 0; f < 6; ++f) slab[f * stride + i] = static_cast<T>(0);
    id[i] = static_cast<uint32_t>(i);
}

constexpr int kBinPer3 = 8;

// the key of the LDS-staged two-level binning (fpic_kernels.hpp); a slot with x < 0 is dead and is not copied
template <typename T, int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
struct BoxTileKey {
    int nx, ny, nz, ntx, nty;
    __device__ __forceinline__ uint32_t operator()(T x, T y, T z) const
    {
        return x < static_cast<T>(0) ? ~0u : key_of<T, LX, LY, LZ>(x, y, z, nx, ny, nz, ntx, nty);
    }
};

template <typename T, int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
__global__ __launch_bounds__(256) void bin3_count_kernel(const T* __restrict__ slab, size_t stride, size_t n, int nx, int ny, int nz, int ntx, int nty,
                                                         uint32_t ntiles, uint32_t* __restrict__ tile_count)
{
    extern __shared__ uint32_t hist3[];
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) hist3[t] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * (256 * kBinPer3);
    for (int k = 0; k < kBinPer3; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        if (i < n && !(slab[i] < static_cast<T>(0))) // (x < 0 marks a slot whose particle has migrated to another rank)
            atomicAdd(&hist3[key_of<T, LX, LY, LZ>(slab[i], slab[stride + i], slab[2 * stride + i], nx, ny, nz, ntx, nty)], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256)
        if (hist3[t]) atomicAdd(&tile_count[t], hist3[t]);
}

// the census of a grid with more tiles than an LDS histogram holds: atomics on the global table, two rounds of
// wave-level aggregation first (an already sorted input sends one atomic per wave instead of 64 to one address)
template <typename T, int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
__global__ __launch_bounds__(256) void bin3_count_global_kernel(const T* __restrict__ slab, size_t stride, size_t n, int nx, int ny, int nz, int ntx, int nty,
                                                                uint32_t* __restrict__ tile_count)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    bool todo = i < n && !(slab[i] < static_cast<T>(0));
    const uint32_t key = todo ? key_of<T, LX, LY, LZ>(slab[i], slab[stride + i], slab[2 * stride + i], nx, ny, nz, ntx, nty) : 0u;
    const int lane = static_cast<int>(threadIdx.x & 63);
    for (int round = 0; round < 2; ++round) {
        const unsigned long long open = __ballot(todo);
        if (!open) return;
        const int leader = __ffsll(open) - 1;
        const uint32_t k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(todo && key == k0);
        if (lane == leader) atomicAdd(&tile_count[k0], static_cast<uint32_t>(__popcll(same)));
        if (key == k0) todo = false;
    }
    if (todo) atomicAdd(&tile_count[key], 1u);
}

// Round 4: the census of a grid with more tiles than an LDS histogram holds WITHOUT a global atomic per particle.  A freshly
// uploaded population is in random order: bin3_count_global_kernel then sends 1e9 atomics to 65 536 counters (41 ms per
// species at configs[3]).  The staged scatter is two-level anyway, so the census is too: (1) coarse bins (div consecutive
// tiles, <= 1024 of them) counted in an LDS histogram; (2) after the coarse pass the array is sorted by coarse bin, and a
// workgroup's chunk lies in one coarse bin (or straddles two): the tiles of that bin are counted in LDS again.
constexpr int kCoarsePer = 8;
template <typename T, typename Key>
__global__ __launch_bounds__(1024) void bin3_count_coarse_kernel(const T* __restrict__ slab, size_t stride, size_t n, Key key_of_pos, uint32_t div, uint32_t ncoarse,
                                                                 uint32_t* __restrict__ coarse_count)
{
    __shared__ uint32_t hist[1024];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * (1024 * kCoarsePer);
#pragma unroll
    for (int k = 0; k < kCoarsePer; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 1024 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t key = key_of_pos(slab[i], slab[stride + i], slab[2 * stride + i]);
        if (key != ~0u) atomicAdd(&hist[key / div], 1u);
    }
    __syncthreads();
    if (threadIdx.x < ncoarse && hist[threadIdx.x]) atomicAdd(&coarse_count[threadIdx.x], hist[threadIdx.x]);
}

// exclusive scan of the coarse counts into the places the coarse pass of sort_scatter_kernel reads (tile_start[c * div]); the
// live total behind the last count
static __global__ __launch_bounds__(1024) void coarse_scan_kernel(uint32_t* __restrict__ coarse_count, uint32_t ncoarse, uint32_t div, uint32_t* __restrict__ tile_start)
{
    __shared__ uint32_t part[1024];
    const uint32_t c = threadIdx.x;
    const uint32_t v = c < ncoarse ? coarse_count[c] : 0;
    part[c] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t t = c >= static_cast<uint32_t>(o) ? part[c - o] : 0;
        __syncthreads();
        part[c] += t;
        __syncthreads();
    }
    if (c < ncoarse) tile_start[c * div] = part[c] - v;
    if (c == 1023) coarse_count[ncoarse] = part[1023];
}

template <typename T, typename Key>
__global__ __launch_bounds__(1024) void bin3_count_sorted_kernel(const T* __restrict__ slab, size_t stride, const uint32_t* __restrict__ live, Key key_of_pos, uint32_t div,
                                                                 uint32_t* __restrict__ tile_count)
{
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t c0s;
    const size_t n = *live, base = static_cast<size_t>(blockIdx.x) * (1024 * kCoarsePer);
    if (base >= n) return;
    hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) c0s = key_of_pos(slab[base], slab[stride + base], slab[2 * stride + base]) / div;
    __syncthreads();
    const uint32_t c0 = c0s;
#pragma unroll
    for (int k = 0; k < kCoarsePer; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 1024 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t key = key_of_pos(slab[i], slab[stride + i], slab[2 * stride + i]);
        if (key == ~0u) continue;                       // (the coarse pass copies no dead slot: cannot happen, costs nothing)
        if (key / div == c0) atomicAdd(&hist[key - c0 * div], 1u);
        else atomicAdd(&tile_count[key], 1u);           // the chunk straddles into the next coarse bin
    }
    __syncthreads();
    if (threadIdx.x < div && hist[threadIdx.x]) atomicAdd(&tile_count[c0 * div + threadIdx.x], hist[threadIdx.x]);
}

template <typename T, int LX = FES_LTX, int LY = FES_LTY, int LZ = FES_LTZ>
__global__ __launch_bounds__(256) void bin3_scatter_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t stride, const uint32_t* __restrict__ src_id,
                                                           uint32_t* __restrict__ dst_id, size_t n, int nx, int ny, int nz, int ntx, int nty,
                                                           uint32_t ntiles, const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor)
{
    extern __shared__ uint32_t hist3[];
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) hist3[t] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * (256 * kBinPer3);
    uint32_t key[kBinPer3], rank[kBinPer3];
#pragma unroll
    for (int k = 0; k < kBinPer3; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        key[k] = ~0u; rank[k] = 0;
        if (i < n && !(src[i] < static_cast<T>(0))) {
            key[k] = key_of<T, LX, LY, LZ>(src[i], src[stride + i], src[2 * stride + i], nx, ny, nz, ntx, nty);
            rank[k] = atomicAdd(&hist3[key[k]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) {
        const uint32_t c = hist3[t];
        if (c) hist3[t] = tile_start[t] + atomicAdd(&tile_cursor[t], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBinPer3; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        if (i >= n || key[k] == ~0u) continue;
        const size_t d = static_cast<size_t>(hist3[key[k]]) + rank[k];
#pragma unroll
        for (int f = 0; f < 6; ++f) dst[f * stride + d] = src[f * stride + i];
        dst_id[d] = src_id[i];
    }
}


// ------------------------------------------------------------------ full EM (Yee FDTD), BASELINE configs[4]
//
// Parity unpinned; defined by oracle/es3d_oracle_impl.h ("full EM") and oracle/es3d_oracle.c (es3d_current) and
// followed operation for operation.  Lattice fields 4 T per node: Ey = (Ex(i+1/2,j,k), Ey(i,j+1/2,k), Ez(i,j,k+1/2), 0),
// By = (Bx(i,j+1/2,k+1/2), By(i+1/2,j,k+1/2), Bz(i+1/2,j+1/2,k), 0); E4n / B4n their node-centred copies; Jfix 3
// int64 per node (x-, y-, z-edge), 96 * 2^42 per particle crossing a whole dual face: with the charge grid the lattice
// continuity equation holds exactly in integers.

#define FES_AT(A, ii, jj, kk, comp) A[4 * (static_cast<size_t>(ii) + sy * static_cast<size_t>(jj) + sz * static_cast<size_t>(kk)) + (comp)]

template <typename T>
__global__ __launch_bounds__(256) void em_nodes_kernel(const T* __restrict__ Ey, const T* __restrict__ By, int nx, int ny, int nz, T* __restrict__ E4n,
                                                       T* __restrict__ B4n, int k0, int nk, Held held)
{
    // planes k0 .. k0 + nk - 1 (periodic): the whole grid for one handle, the slab and its ghost planes for a rank; the
    // arrays hold the planes `held` (the planes named and the one below them are held)
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    if (t >= sz * nk) return;
    const int i = static_cast<int>(t % nx), j = static_cast<int>((t / nx) % ny), kg = (k0 + static_cast<int>(t / sz)) % nz;
    const int k = held_plane(kg, held, nz), km = held_plane(kg ? kg - 1 : nz - 1, held, nz);
    if ((k | km) < 0) return; // (a plane the caller should not have named)
    const size_t c = i + sy * j + sz * k;
    const int im = i ? i - 1 : nx - 1, jm = j ? j - 1 : ny - 1;
    E4n[4 * c] = static_cast<T>(0.5) * (FES_AT(Ey, im, j, k, 0) + FES_AT(Ey, i, j, k, 0));
    E4n[4 * c + 1] = static_cast<T>(0.5) * (FES_AT(Ey, i, jm, k, 1) + FES_AT(Ey, i, j, k, 1));
    E4n[4 * c + 2] = static_cast<T>(0.5) * (FES_AT(Ey, i, j, km, 2) + FES_AT(Ey, i, j, k, 2));
    E4n[4 * c + 3] = static_cast<T>(0);
    B4n[4 * c] = static_cast<T>(0.25) * (((FES_AT(By, i, jm, km, 0) + FES_AT(By, i, j, km, 0)) + FES_AT(By, i, jm, k, 0)) + FES_AT(By, i, j, k, 0));
    B4n[4 * c + 1] = static_cast<T>(0.25) * (((FES_AT(By, im, j, km, 1) + FES_AT(By, i, j, km, 1)) + FES_AT(By, im, j, k, 1)) + FES_AT(By, i, j, k, 1));
    B4n[4 * c + 2] = static_cast<T>(0.25) * (((FES_AT(By, im, jm, k, 2) + FES_AT(By, i, jm, k, 2)) + FES_AT(By, im, j, k, 2)) + FES_AT(By, i, j, k, 2));
    B4n[4 * c + 3] = static_cast<T>(0);
}

// B -= cb * curl E (em_update_b) on the planes k0 .. k0 + nk - 1 (they and the one above them are held)
template <typename T>
__global__ __launch_bounds__(256) void em_update_b_kernel(T* By, const T* __restrict__ Ey, int nx, int ny, int nz, T cbx, T cby, T cbz, int k0,
                                                          int nk, Held held, const T* Bin = nullptr)
{
    if (!Bin) Bin = By; // (the chained step of an undecomposed handle reads one array and writes another)
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    if (t >= sz * nk) return;
    const int i = static_cast<int>(t % nx), j = static_cast<int>((t / nx) % ny), kg = (k0 + static_cast<int>(t / sz)) % nz;
    const int k = held_plane(kg, held, nz), kp = held_plane((kg + 1 == nz) ? 0 : kg + 1, held, nz);
    if ((k | kp) < 0) return;
    const size_t c = i + sy * j + sz * k;
    const int ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1;
    const T cx = (FES_AT(Ey, i, jp, k, 2) - FES_AT(Ey, i, j, k, 2)) * cby - (FES_AT(Ey, i, j, kp, 1) - FES_AT(Ey, i, j, k, 1)) * cbz;
    const T cy = (FES_AT(Ey, i, j, kp, 0) - FES_AT(Ey, i, j, k, 0)) * cbz - (FES_AT(Ey, ip, j, k, 2) - FES_AT(Ey, i, j, k, 2)) * cbx;
    const T cz = (FES_AT(Ey, ip, j, k, 1) - FES_AT(Ey, i, j, k, 1)) * cbx - (FES_AT(Ey, i, jp, k, 0) - FES_AT(Ey, i, j, k, 0)) * cby;
    By[4 * c] = Bin[4 * c] - cx;
    By[4 * c + 1] = Bin[4 * c + 1] - cy;
    By[4 * c + 2] = Bin[4 * c + 2] - cz;
}

// Round 4: the chained lattice step of an undecomposed handle.  The second B half step of one sub-step and the first of the
// next read the same E, and between them only the node centring reads B; em_chain_kernel does all three in ONE sweep: from
// Bh = B at half time and E it forms b1 = Bh - cb curl E — B at the integer time, exactly what em_update_b_kernel would have
// stored — on the face itself and on the three neighbouring faces per component that the node's average takes, writes the
// node-centred E and B (em_nodes_kernel's expressions, operand for operand), and writes b1 - cb curl E, the B at the next
// half time, to the OTHER half-time array (neighbours still read this one).  21.5 GB instead of 43 GB at 512^3 in double.
// Same arithmetic in the same order on every value that is kept: bit-identical to the three separate sweeps.
template <typename T>
__global__ __launch_bounds__(256) void em_chain_kernel(const T* __restrict__ Bh, const T* __restrict__ Ey, int nx, int ny, int nz, T cbx, T cby, T cbz,
                                                       T* __restrict__ E4n, T* __restrict__ B4n, T* __restrict__ Bnext)
{
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    if (t >= sz * nz) return;
    const int i = static_cast<int>(t % nx), j = static_cast<int>((t / nx) % ny), k = static_cast<int>(t / sz);
    const int im = i ? i - 1 : nx - 1, jm = j ? j - 1 : ny - 1, km = k ? k - 1 : nz - 1;
    const int ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1, kp = (k + 1 == nz) ? 0 : k + 1;
    // half a step of curl E at face (a, b, c) whose upper neighbours are (ap, bp, cp): em_update_b_kernel's cx, cy, cz
    auto cx = [&](int a, int b, int c, int bp, int cp) { return (FES_AT(Ey, a, bp, c, 2) - FES_AT(Ey, a, b, c, 2)) * cby - (FES_AT(Ey, a, b, cp, 1) - FES_AT(Ey, a, b, c, 1)) * cbz; };
    auto cy = [&](int a, int b, int c, int ap, int cp) { return (FES_AT(Ey, a, b, cp, 0) - FES_AT(Ey, a, b, c, 0)) * cbz - (FES_AT(Ey, ap, b, c, 2) - FES_AT(Ey, a, b, c, 2)) * cbx; };
    auto cz = [&](int a, int b, int c, int ap, int bp) { return (FES_AT(Ey, ap, b, c, 1) - FES_AT(Ey, a, b, c, 1)) * cbx - (FES_AT(Ey, a, bp, c, 0) - FES_AT(Ey, a, b, c, 0)) * cby; };
    // own face: B at the integer time, then at the next half time
    const T ox = cx(i, j, k, jp, kp), oy = cy(i, j, k, ip, kp), oz = cz(i, j, k, ip, jp);
    const T b1x = FES_AT(Bh, i, j, k, 0) - ox, b1y = FES_AT(Bh, i, j, k, 1) - oy, b1z = FES_AT(Bh, i, j, k, 2) - oz;
    const size_t c = i + sy * j + sz * k;
    Bnext[4 * c] = b1x - ox;
    Bnext[4 * c + 1] = b1y - oy;
    Bnext[4 * c + 2] = b1z - oz;
    Bnext[4 * c + 3] = static_cast<T>(0);
    // the neighbouring faces the node's average takes ((jm)+1 = j, (km)+1 = k, (im)+1 = i)
    const T bx_jm_km = FES_AT(Bh, i, jm, km, 0) - cx(i, jm, km, j, k), bx_j_km = FES_AT(Bh, i, j, km, 0) - cx(i, j, km, jp, k), bx_jm_k = FES_AT(Bh, i, jm, k, 0) - cx(i, jm, k, j, kp);
    const T by_im_km = FES_AT(Bh, im, j, km, 1) - cy(im, j, km, i, k), by_i_km = FES_AT(Bh, i, j, km, 1) - cy(i, j, km, ip, k), by_im_k = FES_AT(Bh, im, j, k, 1) - cy(im, j, k, i, kp);
    const T bz_im_jm = FES_AT(Bh, im, jm, k, 2) - cz(im, jm, k, i, j), bz_i_jm = FES_AT(Bh, i, jm, k, 2) - cz(i, jm, k, ip, j), bz_im_j = FES_AT(Bh, im, j, k, 2) - cz(im, j, k, i, jp);
    E4n[4 * c] = static_cast<T>(0.5) * (FES_AT(Ey, im, j, k, 0) + FES_AT(Ey, i, j, k, 0));
    E4n[4 * c + 1] = static_cast<T>(0.5) * (FES_AT(Ey, i, jm, k, 1) + FES_AT(Ey, i, j, k, 1));
    E4n[4 * c + 2] = static_cast<T>(0.5) * (FES_AT(Ey, i, j, km, 2) + FES_AT(Ey, i, j, k, 2));
    E4n[4 * c + 3] = static_cast<T>(0);
    B4n[4 * c] = static_cast<T>(0.25) * (((bx_jm_km + bx_j_km) + bx_jm_k) + b1x);
    B4n[4 * c + 1] = static_cast<T>(0.25) * (((by_im_km + by_i_km) + by_im_k) + b1y);
    B4n[4 * c + 2] = static_cast<T>(0.25) * (((bz_im_jm + bz_i_jm) + bz_im_j) + b1z);
    B4n[4 * c + 3] = static_cast<T>(0);
}

// The same sweep with the tile's E and half-time B staged in LDS and every face's integer-time B formed ONCE (em_chain_kernel
// forms it up to four times and makes 48 cached loads per node: bound by the L1, profiles/r04_em_chain_ablation.txt).  A
// workgroup owns kCX x kCY x kCZ nodes; it stages E on [-1, T]^3 and Bh on [-1, T-1]^3 (periodic), replaces Bh by b1 in place
// — storing b1 - cb curl E for the faces it owns on the way — and then centres E and b1 on its nodes.  Expressions and operand
// order are em_update_b_kernel's and em_nodes_kernel's.
constexpr int kCX = 32, kCY = 4, kCZ = 4, kChainThreads = 256;   // (49 KB of LDS in double: three workgroups per CU)
template <typename T>
constexpr size_t em_chain_lds_bytes() { return (static_cast<size_t>(kCX + 2) * (kCY + 2) * (kCZ + 2) + static_cast<size_t>(kCX + 1) * (kCY + 1) * (kCZ + 1)) * 3 * sizeof(T); }

template <typename T>
__global__ __launch_bounds__(kChainThreads) void em_chain_tiled_kernel(const T* __restrict__ Bh, const T* __restrict__ Ey, int nx, int ny, int nz, T cbx, T cby, T cbz,
                                                                       T* __restrict__ E4n, T* __restrict__ B4n, T* __restrict__ Bnext, int k0, int nk, Held held,
                                                                       int below_too)
{
    // The node planes k0 .. k0 + nk - 1 (periodic) of the planes `held` holds: the whole lattice of an undecomposed handle
    // (0, nz), or the planes a rank's particles can gather on.  E is read on k0 - 1 .. k0 + nk and the half-time B on
    // k0 - 1 .. k0 + nk - 1; the next half-time B is written on the node planes and, below_too, on plane k0 - 1 as well — a
    // rank forms it on every plane it will read it on, from the E halo it receives, instead of receiving it (fes_api.hip,
    // dom_em_substep).  A plane that is not held reads as zero and is not written.
    extern __shared__ __attribute__((aligned(16))) unsigned char chain_lds[];
    constexpr int EX = kCX + 2, EY = kCY + 2, EZ = kCZ + 2, BX = kCX + 1, BY = kCY + 1, BZ = kCZ + 1;
    T* le = reinterpret_cast<T*>(chain_lds);        // [EZ][EY][EX][3]: E at offsets -1 .. T
    T* lb = le + EX * EY * EZ * 3;                   // [BZ][BY][BX][3]: Bh, then b1, at offsets -1 .. T - 1
    const int tiles_x = (nx + kCX - 1) / kCX, tiles_y = (ny + kCY - 1) / kCY;
    const int x0 = static_cast<int>(blockIdx.x % tiles_x) * kCX, y0 = static_cast<int>((blockIdx.x / tiles_x) % tiles_y) * kCY;
    const int z0 = static_cast<int>(blockIdx.x / (tiles_x * tiles_y)) * kCZ;    // (offset within the range)
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    auto wrap = [](int v, int n) { v %= n; return v < 0 ? v + n : v; };
    auto plane = [&](int c) { return held_plane(wrap(k0 + z0 + c, nz), held, nz); }; // place of the plane at offset c of this tile, or -1
    for (int e = threadIdx.x; e < EX * EY * EZ; e += kChainThreads) {
        const int a = e % EX, b = (e / EX) % EY, c = e / (EX * EY);
        const int pl = plane(c - 1);
        T v0 = 0, v1 = 0, v2 = 0;
        if (pl >= 0) {
            const size_t g = 4 * (static_cast<size_t>(wrap(x0 + a - 1, nx)) + sy * wrap(y0 + b - 1, ny) + sz * pl);
            v0 = Ey[g]; v1 = Ey[g + 1]; v2 = Ey[g + 2];
        }
        le[3 * e] = v0; le[3 * e + 1] = v1; le[3 * e + 2] = v2;
    }
    for (int f = threadIdx.x; f < BX * BY * BZ; f += kChainThreads) {
        const int a = f % BX, b = (f / BX) % BY, c = f / (BX * BY);
        const int pl = plane(c - 1);
        T v0 = 0, v1 = 0, v2 = 0;
        if (pl >= 0) {
            const size_t g = 4 * (static_cast<size_t>(wrap(x0 + a - 1, nx)) + sy * wrap(y0 + b - 1, ny) + sz * pl);
            v0 = Bh[g]; v1 = Bh[g + 1]; v2 = Bh[g + 2];
        }
        lb[3 * f] = v0; lb[3 * f + 1] = v1; lb[3 * f + 2] = v2;
    }
    __syncthreads();
    // E at LDS offsets (a, b, c) in [-1, T], component m
    auto E = [&](int a, int b, int c, int m) { return le[3 * ((a + 1) + EX * ((b + 1) + EY * (c + 1))) + m]; };
    for (int f = threadIdx.x; f < BX * BY * BZ; f += kChainThreads) {
        const int a = f % BX - 1, b = (f / BX) % BY - 1, c = f / (BX * BY) - 1;       // the face's offsets, -1 .. T - 1
        const T cx = (E(a, b + 1, c, 2) - E(a, b, c, 2)) * cby - (E(a, b, c + 1, 1) - E(a, b, c, 1)) * cbz;
        const T cy = (E(a, b, c + 1, 0) - E(a, b, c, 0)) * cbz - (E(a + 1, b, c, 2) - E(a, b, c, 2)) * cbx;
        const T cz = (E(a + 1, b, c, 1) - E(a, b, c, 1)) * cbx - (E(a, b + 1, c, 0) - E(a, b, c, 0)) * cby;
        const T b1x = lb[3 * f] - cx, b1y = lb[3 * f + 1] - cy, b1z = lb[3 * f + 2] - cz;
        lb[3 * f] = b1x; lb[3 * f + 1] = b1y; lb[3 * f + 2] = b1z;
        const int gx = x0 + a, gy = y0 + b, gz = z0 + c;
        // a face of this workgroup's own nodes (or, below_too, of the plane under the range's first)
        const bool mine = a >= 0 && b >= 0 && gx < nx && gy < ny && gz < nk && (c >= 0 || (below_too && gz == -1));
        const int pl = mine ? plane(c) : -1;
        if (pl >= 0) {
            const size_t g = 4 * (static_cast<size_t>(gx) + sy * gy + sz * pl);
            Bnext[g] = b1x - cx; Bnext[g + 1] = b1y - cy; Bnext[g + 2] = b1z - cz; Bnext[g + 3] = static_cast<T>(0);
        }
    }
    __syncthreads();
    auto B1 = [&](int a, int b, int c, int m) { return lb[3 * ((a + 1) + BX * ((b + 1) + BY * (c + 1))) + m]; };
    for (int n = threadIdx.x; n < kCX * kCY * kCZ; n += kChainThreads) {
        const int a = n % kCX, b = (n / kCX) % kCY, c = n / (kCX * kCY);
        const int gx = x0 + a, gy = y0 + b, gz = z0 + c;
        if (gx >= nx || gy >= ny || gz >= nk) continue;
        const int pl = plane(c);
        if (pl < 0) continue;
        const size_t g = 4 * (static_cast<size_t>(gx) + sy * gy + sz * pl);
        E4n[g] = static_cast<T>(0.5) * (E(a - 1, b, c, 0) + E(a, b, c, 0));
        E4n[g + 1] = static_cast<T>(0.5) * (E(a, b - 1, c, 1) + E(a, b, c, 1));
        E4n[g + 2] = static_cast<T>(0.5) * (E(a, b, c - 1, 2) + E(a, b, c, 2));
        E4n[g + 3] = static_cast<T>(0);
        B4n[g] = static_cast<T>(0.25) * (((B1(a, b - 1, c - 1, 0) + B1(a, b, c - 1, 0)) + B1(a, b - 1, c, 0)) + B1(a, b, c, 0));
        B4n[g + 1] = static_cast<T>(0.25) * (((B1(a - 1, b, c - 1, 1) + B1(a, b, c - 1, 1)) + B1(a - 1, b, c, 1)) + B1(a, b, c, 1));
        B4n[g + 2] = static_cast<T>(0.25) * (((B1(a - 1, b - 1, c, 2) + B1(a, b - 1, c, 2)) + B1(a - 1, b, c, 2)) + B1(a, b, c, 2));
        B4n[g + 3] = static_cast<T>(0);
    }
}

// E += ce * curl B - je * J, with J = T((double)Jfix * scale) formed on the fly (em_j_real + em_update_e), on the planes
// k0 .. k0 + nk - 1 (they and the one below them are held)
template <typename T>
__global__ __launch_bounds__(256) void em_update_e_kernel(T* __restrict__ Ey, const T* __restrict__ By, const long long* __restrict__ Jfix, int nx, int ny,
                                                          int nz, T cex, T cey, T cez, T je, double jsx, double jsy, double jsz, int k0, int nk, Held held)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x), j = static_cast<int>(blockIdx.y * blockDim.y + threadIdx.y); // (node_launch)
    if (i >= nx || j >= ny || static_cast<int>(blockIdx.z) >= nk) return;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    const int kg = (k0 + static_cast<int>(blockIdx.z)) % nz;
    const int k = held_plane(kg, held, nz), km = held_plane(kg ? kg - 1 : nz - 1, held, nz);
    if ((k | km) < 0) return;
    const size_t c = i + sy * j + sz * k;
    const int im = i ? i - 1 : nx - 1, jm = j ? j - 1 : ny - 1;
    const T cx = (FES_AT(By, i, j, k, 2) - FES_AT(By, i, jm, k, 2)) * cey - (FES_AT(By, i, j, k, 1) - FES_AT(By, i, j, km, 1)) * cez;
    const T cy = (FES_AT(By, i, j, k, 0) - FES_AT(By, i, j, km, 0)) * cez - (FES_AT(By, i, j, k, 2) - FES_AT(By, im, j, k, 2)) * cex;
    const T cz = (FES_AT(By, i, j, k, 1) - FES_AT(By, im, j, k, 1)) * cex - (FES_AT(By, i, j, k, 0) - FES_AT(By, i, jm, k, 0)) * cey;
    const T jx = static_cast<T>(static_cast<double>(Jfix[3 * c]) * jsx);
    const T jy = static_cast<T>(static_cast<double>(Jfix[3 * c + 1]) * jsy);
    const T jz = static_cast<T>(static_cast<double>(Jfix[3 * c + 2]) * jsz);
    Ey[4 * c] = (Ey[4 * c] + cx) - je * jx;
    Ey[4 * c + 1] = (Ey[4 * c + 1] + cy) - je * jy;
    Ey[4 * c + 2] = (Ey[4 * c + 2] + cz) - je * jz;
}

// E on the Yee edges from phi (em_edge_gradient) on the planes k0 .. k0 + nk - 1 (they and the one above them are held)
template <typename T>
__global__ __launch_bounds__(256) void em_edge_gradient_kernel(const T* __restrict__ phi, int nx, int ny, int nz, T hx, T hy, T hz, T* __restrict__ Ey, int k0, int nk,
                                                               Held held)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x), j = static_cast<int>(blockIdx.y * blockDim.y + threadIdx.y); // (node_launch)
    if (i >= nx || j >= ny || static_cast<int>(blockIdx.z) >= nk) return;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    int kg = (k0 + static_cast<int>(blockIdx.z)) % nz;
    if (kg < 0) kg += nz;
    const int k = held_plane(kg, held, nz), kp = held_plane((kg + 1 == nz) ? 0 : kg + 1, held, nz);
    if ((k | kp) < 0) return;
    const size_t c = i + sy * j + sz * k;
    const int ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1;
    Ey[4 * c] = (phi[c] - phi[ip + sy * j + sz * k]) * hx;
    Ey[4 * c + 1] = (phi[c] - phi[i + sy * jp + sz * k]) * hy;
    Ey[4 * c + 2] = (phi[c] - phi[i + sy * j + sz * kp]) * hz;
    Ey[4 * c + 3] = static_cast<T>(0);
}
#undef FES_AT

template <typename T>
__global__ __launch_bounds__(256) void fill4_kernel(T* __restrict__ a, size_t nodes, T x, T y, T z)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= nodes) return;
    a[4 * c] = x; a[4 * c + 1] = y; a[4 * c + 2] = z; a[4 * c + 3] = static_cast<T>(0);
}

// doubled fixed-point lattice coordinate (em_coord)
template <typename T>
__device__ __forceinline__ long long em_coord(T u, int n)
{
    int i, w1;
    axis(u, n, i, w1);
    return 2 * (static_cast<long long>(i) * 16384 + w1);
}

__device__ __forceinline__ long long floor_div_ll(long long a, long long s) { return a >= 0 ? a / s : -((-a + s - 1) / s); }

// one straight segment inside one cell (current_segment).  Not inlined: the rare global path must not bloat the
// unrolled particle loops of its callers.
__device__ __attribute__((noinline)) void current_segment(const long long (&p1)[3], const long long (&p2)[3], const long long (&cell)[3], int nx, int ny, int nz, int Z,
                                                unsigned long long* Jfix, Held held)
{
    constexpr long long S = 32768;
    long long d[3], A0[3], A1[3], c0[3], c1[3];
    const int n[3] = { nx, ny, nz };
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long l1 = p1[m] - cell[m] * S, l2 = p2[m] - cell[m] * S;
        d[m] = l2 - l1;
        A1[m] = l1 + l2; A0[m] = 2 * S - A1[m];
        long long w = cell[m] % n[m];
        if (w < 0) w += n[m];
        c0[m] = w;
        c1[m] = (w + 1 == n[m]) ? 0 : w + 1;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (d[m] == 0) continue;
        const int u = (m + 1) % 3, v = (m + 2) % 3;
        const long long cross = d[u] * d[v];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long flux = d[m] * (3 * (b ? A1[u] : A0[u]) * (c ? A1[v] : A0[v]) + (b == c ? cross : -cross)) * Z;
                long long idx[3];
                idx[m] = c0[m]; idx[u] = b ? c1[u] : c0[u]; idx[v] = c ? c1[v] : c0[v];
                const int lk = held_plane(static_cast<int>(idx[2]), held, nz);
                if (flux && lk >= 0) atomicAdd(Jfix + 3 * (static_cast<size_t>(idx[0]) + static_cast<size_t>(nx) * (static_cast<size_t>(idx[1]) + static_cast<size_t>(ny) * lk)) + m,
                                               static_cast<unsigned long long>(flux));
            }
    }
}

// the move a -> b_in (nearest periodic image) cut at the zigzag relay point (es3d_current)
__device__ __forceinline__ void current_deposit(const long long (&a)[3], const long long (&b_in)[3], int nx, int ny, int nz, int Z, unsigned long long* Jfix, Held held)
{
    constexpr long long S = 32768;
    const int n[3] = { nx, ny, nz };
    long long b[3], ca[3], cb[3], r[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long box = static_cast<long long>(n[m]) * S;
        long long dd = b_in[m] - a[m];
        if (2 * dd > box) dd -= box;
        else if (2 * dd < -box) dd += box;
        b[m] = a[m] + dd;
        ca[m] = floor_div_ll(a[m], S);
        cb[m] = floor_div_ll(b[m], S);
        r[m] = (ca[m] == cb[m]) ? (a[m] + b[m]) / 2 : (ca[m] > cb[m] ? ca[m] : cb[m]) * S;
    }
    current_segment(a, r, ca, nx, ny, nz, Z, Jfix, held);
    current_segment(r, b, cb, nx, ny, nz, Z, Jfix, held);
}

template <typename T>
struct EmPushArgs {
    T* slab;
    size_t stride;
    unsigned long long n;
    const T* E4n;
    const T* B4n;
    unsigned long long* Jfix;
    int nx, ny, nz;
    Held held;               // the planes E4n, B4n and Jfix hold
    T h, hc, dx, dy, dz;
    int Z;
};

// em_push after the gather: Boris with the particle's own t = hB, s = 2t / (1 + t^2), drift, wrap
template <typename T>
__device__ __forceinline__ void em_boris_move(P3<T>& p, T E0, T E1, T E2, T B0, T B1, T B2, T h, T hc, T dx, T dy, T dz)
{
    const T ax = hc * E0, ay = hc * E1, az = hc * E2;
    const T tx = h * B0, ty = h * B1, tz = h * B2;
    const T f = static_cast<T>(2) / (static_cast<T>(1) + fma_(tz, tz, fma_(ty, ty, tx * tx)));
    const T sx = f * tx, sy = f * ty, sz = f * tz;
    const T ux = p.vx + ax, uy = p.vy + ay, uz = p.vz + az;
    const T px = ux + fma_(uy, tz, -(uz * ty));
    const T py = uy + fma_(uz, tx, -(ux * tz));
    const T pz = uz + fma_(ux, ty, -(uy * tx));
    const T qx = ux + fma_(py, sz, -(pz * sy));
    const T qy = uy + fma_(pz, sx, -(px * sz));
    const T qz = uz + fma_(px, sy, -(py * sx));
    p.vx = qx + ax; p.vy = qy + ay; p.vz = qz + az;
    p.x = wrap01(fma_(dx, p.vx, p.x));
    p.y = wrap01(fma_(dy, p.vy, p.y));
    p.z = wrap01(fma_(dz, p.vz, p.z));
}

// em_push + es3d_current of ONE particle against global memory: gathers through L2, the current with 8-byte global
// atomics.  The body of the flat kernel, and the out-of-line rare path of the tiled kernel.
template <typename T, bool LEAN = false>
__device__ __forceinline__ void em_particle_global(P3<T>& p, const T* __restrict__ E4n, const T* __restrict__ B4n, unsigned long long* Jfix, int nx, int ny, int nz, T h, T hc,
                                                   T dx, T dy, T dz, int Z, Held held)
{
    int i, j, k, w1;
    T fx[2], fy[2], fz[2];
    axis(p.x, nx, i, w1); weights_of(w1, fx);
    const long long hx0 = 2 * (static_cast<long long>(i) * 16384 + w1);
    axis(p.y, ny, j, w1); weights_of(w1, fy);
    const long long hy0 = 2 * (static_cast<long long>(j) * 16384 + w1);
    axis(p.z, nz, k, w1); weights_of(w1, fz);
    const long long hz0 = 2 * (static_cast<long long>(k) * 16384 + w1);
    T E[3] = { 0, 0, 0 }, B[3] = { 0, 0, 0 };
    // (LEAN: the tiled kernel's out-of-line copy, rolled up: the registers a callee uses are registers its caller must
    // save around the call, and the caller is the hot loop)
#pragma unroll LEAN ? 1 : 2
    for (int c = 0; c < 2; ++c)
#pragma unroll LEAN ? 1 : 2
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const int ii = (i + aa == nx) ? 0 : i + aa, jj = (j + b == ny) ? 0 : j + b, kk = held_plane((k + c == nz) ? 0 : k + c, held, nz);
                const size_t node = static_cast<size_t>(ii) + static_cast<size_t>(nx) * (static_cast<size_t>(jj) + static_cast<size_t>(ny) * (kk < 0 ? 0 : kk));
                T e[4] = { 0, 0, 0, 0 }, bb[4] = { 0, 0, 0, 0 };
                if (kk >= 0) {
                    fpic::load4(E4n + 4 * node, e);
                    fpic::load4(B4n + 4 * node, bb);
                }
                const T w = (fx[aa] * fy[b]) * fz[c];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    E[m] = fma_(w, e[m], E[m]);
                    B[m] = fma_(w, bb[m], B[m]);
                }
            }
    em_boris_move<T>(p, E[0], E[1], E[2], B[0], B[1], B[2], h, hc, dx, dy, dz);
    const long long from[3] = { hx0, hy0, hz0 };
    const long long to[3] = { em_coord(p.x, nx), em_coord(p.y, ny), em_coord(p.z, nz) };
    current_deposit(from, to, nx, ny, nz, Z, Jfix, held);
}

// Flat form (any particle order; used until the particles have been binned)
template <typename T>
__global__ __launch_bounds__(256) void em_push_kernel(EmPushArgs<T> a)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= a.n) return;
    P3<T> q;
    q.x = a.slab[p]; q.y = a.slab[a.stride + p]; q.z = a.slab[2 * a.stride + p];
    q.vx = a.slab[3 * a.stride + p]; q.vy = a.slab[4 * a.stride + p]; q.vz = a.slab[5 * a.stride + p];
    if (q.x < static_cast<T>(0)) return; // a migrated slot
    em_particle_global<T>(q, a.E4n, a.B4n, a.Jfix, a.nx, a.ny, a.nz, a.h, a.hc, a.dx, a.dy, a.dz, a.Z, a.held);
    a.slab[p] = q.x; a.slab[a.stride + p] = q.y; a.slab[2 * a.stride + p] = q.z;
    a.slab[3 * a.stride + p] = q.vx; a.slab[4 * a.stride + p] = q.vy; a.slab[5 * a.stride + p] = q.vz;
}

// Tiled form of the full-EM push: 8x8x8-cell tiles, so that both node-centred fields (6 T per node) and the three
// int64 current accumulators of a tile and its halo fit in LDS (11^3 nodes x 48 B = 64 KB float, x 72 B = 96 KB double).
// A particle's gather is 24 LDS reads of two T each, its current 12 ds_add_u64 when it stays in its cell (the two
// half-segments of es3d_current are merged: their sum equals the whole segment's fluxes exactly).  Everything else — a
// face crossing, a cell outside the window, a weight rounded up to a whole cell — is an out-of-line call with the
// particle's numbers BY VALUE (and the grid through a resident copy of the arguments), so that the common path keeps no
// state in memory.
#if defined(FES_ABL_EM)                 // development probes (timing only): 4 = no current deposit in the common case, 8 = no window staging / flush, 16 = out-of-window particles skipped
#define FES_ABL_EM_V FES_ABL_EM
#else
#define FES_ABL_EM_V 0
#endif
// Round 5 ("two workgroups per CU for the 3-D pushes ... or the same-box ablation that shows why not";
// profiles/r05_em_pipe_ablation.txt).  The window's records are UNPADDED AND INTERLEAVED — (E0 E1 E2 B0 B1 B2), 6 T per
// node, read as three 2 T vectors: double three ds_read_b128 where the padded pair of records took four instructions, float
// three ds_read_b64 feeding three v_pk_fma_f32 where the padded pair fed four — 11^3 nodes x (6 T + 24 B) = 64 KB float /
// 96 KB double instead of 74.5 / 117.  That alone took the double-precision push from 30.9 to 26.2 ms at 512^3 / 1e9 (7.45
// per cell) and from 11.6 to 11.1 ms at 256^3 / 5e8.  Two further forms were built on it, measured on one box and NOT kept
// as the default:
//   B  an 8 x 4 x 8-cell tile in double (11 x 7 x 11 nodes x 72 B = 61 KB), two workgroups of 256 threads per CU (the form the
//      verdict asked for): 11.2 / 26.7 ms — a tie with one workgroup of 768 on the 8^3 tile (11.1 / 26.2): with 166 VGPRs a
//      SIMD holds three waves, so two workgroups are 8 waves where one is 12, and twice the tiles are twice the windows;
//   A  FES_EM_PIPE=1: a PERSISTENT workgroup of 768 threads (all twelve waves) that walks the work list with TWO windows —
//      while its waves push item k out of window k & 1 each of them first flushes item k - 1's currents from the other
//      window and stages item k + 1's fields into it, no barrier in between, ONE barrier per item.  Two windows fit only
//      with the 8 x 4 x 8 tile in double (2 x 61 KB), and a tile of 256 cells holds 1 900 particles at 7.45 per cell: 950
//      groups on 768 threads are 1.24 turns, the second a quarter full — 11.8 / 35.1 ms.  In float (8^3 tile, 2 x 64 KB,
//      768 threads) it runs like two workgroups of 256: 7.46 / 19.2 against 7.47 / 19.8 ms, both 4-14 % ahead of one
//      workgroup of 768 with one window (7.74 / 22.4).
//   E  FES_EM_PIPE=2: persistent, 8^3 tile, only the FIELD window double-buffered (double: 2 x 64 + 32 KB; the next item's records
//      are staged while this item's particles are pushed; push - barrier - flush - barrier per item): on one box, alternating
//      with the default, 6 % SLOWER in double (10.74 / 26.8 against 10.07 / 25.3 ms), 2 % faster in float.
// So what a second resident tile buys is what the float kernel already had (two workgroups of 256), the double kernel cannot
// have it without halving its tile, and halving the tile costs what it buys.  The default: one item per workgroup, one
// window; float 2 x 256 threads, double 1 x 768.  Bit-identical results in every form (the whole EM suite ran on A).
#if !defined(FES_EM_PIPE)
#define FES_EM_PIPE 0
#endif
#if !defined(FES_EM_LY_F32)
#define FES_EM_LY_F32 3                // log2 of the tile's extent along y
#endif
#if !defined(FES_EM_LY_F64)
#define FES_EM_LY_F64 (FES_EM_PIPE == 1 ? 2 : 3)
#endif
#if !defined(FES_EM_THREADS_F32)
#define FES_EM_THREADS_F32 (FES_EM_PIPE ? 768 : 256)
#endif
#if !defined(FES_EM_THREADS_F64)
#define FES_EM_THREADS_F64 768
#endif
#if !defined(FES_EM_WAVES)
#define FES_EM_WAVES 3          // waves per SIMD the register allocation aims at
#endif
constexpr int kEL = 3;                 // log2 of the EM tile's extent along x and z
template <typename T>
constexpr int em_threads() { return sizeof(T) == 4 ? FES_EM_THREADS_F32 : FES_EM_THREADS_F64; }
template <typename T>
struct EmWin {
    static constexpr int LX = kEL, LY = sizeof(T) == 4 ? FES_EM_LY_F32 : FES_EM_LY_F64, LZ = kEL;
    static constexpr int TX = 1 << LX, TY = 1 << LY, TZ = 1 << LZ;
    // one halo cell: an EM step moves a particle by a small fraction of a cell (c dt < dx / sqrt 3)
    static constexpr int H = 1;
    static constexpr int WX = TX + 2 * H + 1, WY = TY + 2 * H + 1, WZ = TZ + 2 * H + 1;
    static constexpr int N = WX * WY * WZ;
    static constexpr int BUFFERS = FES_EM_PIPE ? 2 : 1;
    static constexpr size_t kFieldBytes = static_cast<size_t>(N) * 6 * sizeof(T);
    static constexpr size_t kBufBytes = (kFieldBytes + static_cast<size_t>(N) * 24 + 15) / 16 * 16;
    static_assert(kFieldBytes % 8 == 0 && kBufBytes % 16 == 0, "the accumulators behind the records are 8-byte aligned, a window's records as aligned as a pair of T");
};
// FES_EM_PIPE == 2: two FIELD windows and ONE accumulator window (double, 8^3 tile: 2 x 64 KB + 32 KB = 159.7 KB)
template <typename T>
constexpr size_t em_lds_bytes()
{
    return FES_EM_PIPE == 2 ? 2 * EmWin<T>::kFieldBytes + static_cast<size_t>(EmWin<T>::N) * 24 + 16 : EmWin<T>::BUFFERS * EmWin<T>::kBufBytes + 16;
}

template <typename T>
struct EmTileArgs {
    EmPushArgs<T> p;
    int ntx, nty, ntz;
    const BlockWork* work;
    const uint32_t* nwork;
    int part;                // see Push3Args
    uint32_t tiles_per_layer, layer_lo, layer_hi;
    unsigned long long* spilled;
    const uint32_t* tile_start;    // the live bin table (the slots of the interior layers: fes_groups.hpp)
    const EmPushArgs<T>* resident; // a copy of p in device memory: what the out-of-line rare paths read their grid from
};

template <typename A>
__global__ void store_args_kernel(A a, A* dst) { *dst = a; }

// window slot of cell (i,j,k) (all eight corner nodes inside), or -1; cells lie within one box length of the grid
template <typename T>
__device__ __forceinline__ int em_slot(int i, int j, int k, int ox, int oy, int oz, int nx, int ny, int nz)
{
    const unsigned l = wrap_near(i - ox, nx), m = wrap_near(j - oy, ny), n = wrap_near(k - oz, nz);
    constexpr unsigned WX = EmWin<T>::WX, WY = EmWin<T>::WY, WZ = EmWin<T>::WZ;
    const bool in = l <= WX - 2 && m <= WY - 2 && n <= WZ - 2;
    return in ? static_cast<int>(__umul24(__umul24(n, WY) + m, WX) + l) : -1;
}

// fluxes of one straight segment inside the cell whose window slot is s (current_segment, LDS accumulators)
template <typename T>
__device__ __forceinline__ void current_cell(const int (&p1)[3], const int (&p2)[3], const int (&cell)[3], int s, int Z, FPIC_LDS unsigned long long* lJ)
{
    constexpr int S = 32768;
    const int step[3] = { 1, EmWin<T>::WX, EmWin<T>::WX * EmWin<T>::WY };
    long long d[3], A0[3], A1[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long long l1 = p1[m] - cell[m] * S, l2 = p2[m] - cell[m] * S;
        d[m] = l2 - l1;
        A1[m] = l1 + l2; A0[m] = 2 * S - A1[m];
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (d[m] == 0) continue;
        const int u = (m + 1) % 3, v = (m + 2) % 3;
        const long long cross = d[u] * d[v];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long flux = d[m] * (3 * (b ? A1[u] : A0[u]) * (c ? A1[v] : A0[v]) + (b == c ? cross : -cross)) * Z;
                __hip_atomic_fetch_add(lJ + 3 * (s + b * step[u] + c * step[v]) + m, static_cast<unsigned long long>(flux), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    }
}

// The common case — the particle stays in its cell — in 32-bit arithmetic.  In single (not doubled) units g = H/2 the
// whole-segment flux along m is 8 Z d_m [3 s_u s_v +- d_u d_v] with s = g1 + g2 <= 2^15, |d| < 2^14: the bracket is a
// non-negative number below 2^32 formed by ONE 24-bit multiply-add ((3 s_u) s_v + cross), and one 32 x 32 -> 64
// multiply-add finishes it: for a negative factor f = 8 Z d_m, taken as the unsigned word f + 2^32,
// f * t = (f + 2^32) * t - 2^32 * t, i.e. the addend's high word is -t (generic 64-bit products and the conditional
// negation were most of this kernel's VALU work).  Equal to the oracle's two half-segments: all terms are integers.
// g1 = (wx, wy, wz): the start inside the cell, d: the move, s: the cell's window slot.  A zero move along m adds
// zeros (no branch: it is rare for a thermal particle and the atomics issue for the wave anyway).
template <typename T>
__device__ __forceinline__ void current_cell_fast(int wx, int wy, int wz, int d0, int d1, int d2, int s, int Z, FPIC_LDS unsigned long long* lJ)
{
    const int step[3] = { 1, EmWin<T>::WX, EmWin<T>::WX * EmWin<T>::WY };
    const int d[3] = { d0, d1, d2 };
    const unsigned A1[3] = { static_cast<unsigned>(2 * wx + d0), static_cast<unsigned>(2 * wy + d1), static_cast<unsigned>(2 * wz + d2) };
    unsigned A[3][2], A3[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        A[m][1] = A1[m]; A[m][0] = 32768u - A1[m];
        A3[m][0] = 3u * A[m][0]; A3[m][1] = 3u * A[m][1];
    }
    FPIC_LDS unsigned long long* cell = lJ + 3 * s;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int u = (m + 1) % 3, v = (m + 2) % 3;
        const unsigned cross = static_cast<unsigned>(__mul24(d[u], d[v]));
        const unsigned f = static_cast<unsigned>(__mul24(8 * Z, d[m]));
        const unsigned neg = static_cast<unsigned>(static_cast<int>(f) >> 31); // all ones for a negative factor
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned term = __umul24(A3[u][b], A[v][c]) + (b == c ? cross : 0u - cross);
                const unsigned long long flux = static_cast<unsigned long long>(f) * term + (static_cast<unsigned long long>((0u - term) & neg) << 32);
                __hip_atomic_fetch_add(cell + 3 * (b * step[u] + c * step[v]) + m, flux, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    }
}

// The rare moves of the tiled kernel, out of line, by value: from (f*) to (t*, nearest image) in single fixed-point
// units (cell * 2^14 + weight).  es3d_current in doubled coordinates: the relay point, one segment per cell, each into
// the LDS window when its cell lies inside and into global memory otherwise (through g, the resident copy of the
// arguments: a face crossing is one particle in ten and should not wait for a load; this is none).  Returns the number
// of segments that went to global memory.
template <typename T>
__device__ __attribute__((noinline)) unsigned em_current_rare(int f0, int f1, int f2, int t0, int t1, int t2, int ox, int oy, int oz, int nx, int ny, int nz, int Z,
                                                              FPIC_LDS unsigned long long* lJ, const EmPushArgs<T>* g)
{
    constexpr int S = 32768;
    const int from[3] = { 2 * f0, 2 * f1, 2 * f2 }, to[3] = { 2 * t0, 2 * t1, 2 * t2 };
    int ca[3], cb[3], r[3];
    bool same = true;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        ca[m] = from[m] >> 15;   // floor division by S (arithmetic shift)
        cb[m] = to[m] >> 15;
        same &= ca[m] == cb[m];
        r[m] = (ca[m] == cb[m]) ? (from[m] + to[m]) / 2 : (ca[m] > cb[m] ? ca[m] : cb[m]) * S;
    }
    unsigned spilled = 0;
    auto segment = [&](const int (&p1)[3], const int (&p2)[3], const int (&cell)[3]) {
        const int s = em_slot<T>(cell[0], cell[1], cell[2], ox, oy, oz, nx, ny, nz);
        if (s >= 0) {
            current_cell<T>(p1, p2, cell, s, Z, lJ);
        } else {
            const long long q1[3] = { p1[0], p1[1], p1[2] }, q2[3] = { p2[0], p2[1], p2[2] }, cc[3] = { cell[0], cell[1], cell[2] };
            current_segment(q1, q2, cc, nx, ny, nz, Z, g->Jfix, g->held); // (a face crossing is one particle in ten; this, none)
            ++spilled;
        }
    };
    if (same) { // one cell (outside the window, or reached by a weight that rounded up to a whole cell): the whole segment at once
        segment(from, to, ca);
    } else {
        segment(from, r, ca);
        segment(r, to, cb);
    }
    return spilled;
}

// a particle whose cell is not inside the window: the whole sub-step against global memory, out of line
template <typename T>
__device__ __attribute__((noinline)) P3<T> em_particle_rare(P3<T> p, const EmPushArgs<T>* g)
{
    em_particle_global<T, true>(p, g->E4n, g->B4n, g->Jfix, g->nx, g->ny, g->nz, g->h, g->hc, g->dx, g->dy, g->dz, g->Z, g->held);
    return p;
}

// two T as one LDS access (ds_read/write_b64 in float, _b128 in double)
template <typename T> struct Pair2;
template <> struct Pair2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair2<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T>
__device__ __forceinline__ void em_tile_origin(const EmTileArgs<T>& t, uint32_t tile, int& ox, int& oy, int& oz)
{
    const int ti = static_cast<int>(tile % t.ntx), tj = static_cast<int>((tile / t.ntx) % t.nty), tk = static_cast<int>(tile / (t.ntx * t.nty));
    ox = ti * EmWin<T>::TX - EmWin<T>::H; oy = tj * EmWin<T>::TY - EmWin<T>::H; oz = tk * EmWin<T>::TZ - EmWin<T>::H;
}

// the node-centred fields of a tile's window into LDS: one record (E0 E1 E2 B0 B1 B2) per node; ZERO: the window's current
// accumulators are cleared on the way (the one-window form; the pipelined form's flush leaves them clear)
template <typename T, int THREADS, bool ZERO>
__device__ __forceinline__ void em_stage(const EmPushArgs<T>& a, int ox, int oy, int oz, FPIC_LDS T* lF, FPIC_LDS unsigned long long* lJ)
{
    using P2 = typename Pair2<T>::type;
    constexpr int WX = EmWin<T>::WX, WY = EmWin<T>::WY, WN = EmWin<T>::N;
    const Held hs = a.held;
    for (int s = threadIdx.x; s < ((FES_ABL_EM_V & 8) ? 0 : WN); s += THREADS) {
        const int n = s / (WX * WY), rem = s - n * (WX * WY);
        const int m = rem / WX, l = rem - m * WX;
        // (wrap_window: no division by a run-time number for boxes of 32 nodes or more)
        const int gi = wrap_window(ox + l, a.nx), gj = wrap_window(oy + m, a.ny),
                  gk = wrap_window(oz - hs.zs0 + n, a.nz); // gk: among the planes held
        const int lk = gk < hs.nzs ? gk : -1;
        const size_t node = static_cast<size_t>(gi) + static_cast<size_t>(a.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(a.ny) * (lk < 0 ? 0 : lk));
        if constexpr (ZERO) { lJ[3 * s] = 0ull; lJ[3 * s + 1] = 0ull; lJ[3 * s + 2] = 0ull; }
        T e[4] = { 0, 0, 0, 0 }, b[4] = { 0, 0, 0, 0 }; // (a plane the rank does not hold: no particle of the tile gathers there)
        if (lk >= 0) {
            fpic::load4(a.E4n + 4 * node, e);
            fpic::load4(a.B4n + 4 * node, b);
        }
        FPIC_LDS P2* rec = reinterpret_cast<FPIC_LDS P2*>(lF + 6 * s);
        rec[0] = P2{ e[0], e[1] }; rec[1] = P2{ e[2], b[0] }; rec[2] = P2{ b[1], b[2] };
    }
}

// the window's currents onto the grid (8-byte global atomics, the non-zero ones); the accumulators are left clear
template <typename T, int THREADS>
__device__ __forceinline__ void em_flush(const EmPushArgs<T>& a, int ox, int oy, int oz, FPIC_LDS unsigned long long* lJ)
{
    constexpr int WX = EmWin<T>::WX, WY = EmWin<T>::WY, WN = EmWin<T>::N;
    const Held hf = a.held;
    for (int s3 = threadIdx.x; s3 < ((FES_ABL_EM_V & 8) ? 0 : 3 * WN); s3 += THREADS) {
        const unsigned long long val = lJ[s3];
        if (val == 0ull) continue;
        lJ[s3] = 0ull;
        const int s = s3 / 3, m3 = s3 - 3 * s;
        const int n = s / (WX * WY), rem = s - n * (WX * WY);
        const int m = rem / WX, l = rem - m * WX;
        const int gi = wrap_window(ox + l, a.nx), gj = wrap_window(oy + m, a.ny),
                  gk = wrap_window(oz - hf.zs0 + n, a.nz);
        if (gk < hf.nzs) atomicAdd(a.Jfix + 3 * (static_cast<size_t>(gi) + static_cast<size_t>(a.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(a.ny) * gk)) + m3, val);
    }
}

// the particles of one work item against one window; v holds the lane's first group (asked for by the caller before the
// window work, so that the two trips to memory overlap), g its index
template <typename T, int THREADS>
__device__ __forceinline__ void em_push_item(const EmTileArgs<T>& t, const BlockWork w, int ox, int oy, int oz, const FPIC_LDS T* lF, FPIC_LDS unsigned long long* lJ,
                                             T (&v)[6][Vec16<T>::N], size_t g, size_t g_end, unsigned& my_spill)
{
    using P2 = typename Pair2<T>::type;
    const EmPushArgs<T>& a = t.p;
    constexpr int PPT = Vec16<T>::N;
    constexpr int WX = EmWin<T>::WX, WY = EmWin<T>::WY;
    while (g < g_end) {
        const size_t base = g * PPT;
        const bool whole = base >= w.begin && base + PPT <= w.end; // (w.end <= n)
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!whole && !fesgrp::owns(w.begin, w.end, base + q)) continue;
            T x = v[0][q], y = v[1][q], z = v[2][q];
            if (x < static_cast<T>(0)) continue;
            // cell, upper weight (14-bit fixed point) and single fixed-point coordinate cell * 2^14 + weight per axis
            int i, j, k, wx1, wy1, wz1;
            axis(x, a.nx, i, wx1); axis(y, a.ny, j, wy1); axis(z, a.nz, k, wz1);
            const int s0 = em_slot<T>(i, j, k, ox, oy, oz, a.nx, a.ny, a.nz);
            if (s0 < 0) { // rare: the cell has left the window
                if constexpr ((FES_ABL_EM_V & 16) != 0) continue; // (timing probe: what the out-of-window particles cost)
                P3<T> p{ x, y, z, v[3][q], v[4][q], v[5][q] };
                p = em_particle_rare<T>(p, t.resident);
                v[0][q] = p.x; v[1][q] = p.y; v[2][q] = p.z; v[3][q] = p.vx; v[4][q] = p.vy; v[5][q] = p.vz;
                ++my_spill;
                continue;
            }
            T fx[2], fy[2], fz[2];
            weights_of(wx1, fx); weights_of(wy1, fy); weights_of(wz1, fz);
            T E[3], B[3];
            if constexpr (sizeof(T) == 4) {
                // float: (E0, E1), (E2, B0), (B1, B2) are accumulated by one v_pk_fma_f32 each
                f32x2 e01 = { 0.f, 0.f }, e2b0 = { 0.f, 0.f }, b12 = { 0.f, 0.f };
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa) {
                            const FPIC_LDS P2* rec = reinterpret_cast<const FPIC_LDS P2*>(lF + 6 * (s0 + aa + WX * b + WX * WY * c));
                            const P2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                            const T wgt = (fx[aa] * fy[b]) * fz[c];
                            e01 = fma2_(wgt, f32x2{ r0.x, r0.y }, e01);
                            e2b0 = fma2_(wgt, f32x2{ r1.x, r1.y }, e2b0);
                            b12 = fma2_(wgt, f32x2{ r2.x, r2.y }, b12);
                        }
                E[0] = e01.x; E[1] = e01.y; E[2] = e2b0.x; B[0] = e2b0.y; B[1] = b12.x; B[2] = b12.y;
            } else {
                E[0] = E[1] = E[2] = B[0] = B[1] = B[2] = static_cast<T>(0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa) {
                            const FPIC_LDS P2* rec = reinterpret_cast<const FPIC_LDS P2*>(lF + 6 * (s0 + aa + WX * b + WX * WY * c));
                            const P2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                            const T wgt = (fx[aa] * fy[b]) * fz[c];
                            E[0] = fma_(wgt, r0.x, E[0]); E[1] = fma_(wgt, r0.y, E[1]); E[2] = fma_(wgt, r1.x, E[2]);
                            B[0] = fma_(wgt, r1.y, B[0]); B[1] = fma_(wgt, r2.x, B[1]); B[2] = fma_(wgt, r2.y, B[2]);
                        }
            }
            {
                P3<T> p{ x, y, z, v[3][q], v[4][q], v[5][q] };
                em_boris_move<T>(p, E[0], E[1], E[2], B[0], B[1], B[2], a.h, a.hc, a.dx, a.dy, a.dz);
                x = p.x; y = p.y; z = p.z;
                v[0][q] = x; v[1][q] = y; v[2][q] = z; v[3][q] = p.vx; v[4][q] = p.vy; v[5][q] = p.vz;
            }
            // current (es3d_current), in single fixed-point units: the move d = to - from (nearest periodic image)
            int d0, d1, d2;
            {
                int ci, cw;
                axis(x, a.nx, ci, cw); d0 = ((ci - i) << 14) + (cw - wx1);
                axis(y, a.ny, ci, cw); d1 = ((ci - j) << 14) + (cw - wy1);
                axis(z, a.nz, ci, cw); d2 = ((ci - k) << 14) + (cw - wz1);
                const int hx = a.nx << 13, hy = a.ny << 13, hz = a.nz << 13; // half a box
                d0 += d0 > hx ? -2 * hx : (d0 < -hx ? 2 * hx : 0);
                d1 += d1 > hy ? -2 * hy : (d1 < -hy ? 2 * hy : 0);
                d2 += d2 > hz ? -2 * hz : (d2 < -hz ? 2 * hz : 0);
            }
            // the common case: the particle stays in the cell it was gathered in (no weight was rounded up to a whole
            // cell), i.e. 0 <= w1 + d < 2^14 on every axis
            const int e0 = wx1 + d0, e1 = wy1 + d1, e2 = wz1 + d2;
            const bool common = ((e0 | e1 | e2) & ~16383) == 0 && ((wx1 | wy1 | wz1) & 16384) == 0;
#if defined(FES_ABL_EM) && (FES_ABL_EM & 4)       // development probe (timing only): no current deposit in the common case
            if (common) { if (d0 == 0x7fffffff) lJ[s0] = 1ull; }
#else
            if (common) current_cell_fast<T>(wx1, wy1, wz1, d0, d1, d2, s0, a.Z, lJ);
#endif
            else
                my_spill += em_current_rare<T>((i << 14) + wx1, (j << 14) + wy1, (k << 14) + wz1, (i << 14) + e0, (j << 14) + e1, (k << 14) + e2, ox, oy, oz, a.nx, a.ny,
                                               a.nz, a.Z, lJ, t.resident);
        }
        if (whole) {
#pragma unroll
            for (int f = 0; f < 6; ++f) store_lane<T, PPT>(a.slab + f * a.stride, base, v[f]);
        } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                if (fesgrp::owns(w.begin, w.end, base + q)) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) a.slab[f * a.stride + base + q] = v[f][q];
                }
        }
        g += THREADS;
        if (g < g_end) {
#pragma unroll
            for (int f = 0; f < 6; ++f) load_lane<T, PPT>(a.slab + f * a.stride, g * PPT, v[f]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(em_threads<T>()) __attribute__((amdgpu_waves_per_eu(FES_EM_WAVES))) void em_push_tiles_kernel(EmTileArgs<T> t)
{
    constexpr int kEmThreads = em_threads<T>();
    const EmPushArgs<T>& a = t.p;
    constexpr int PPT = Vec16<T>::N;
    constexpr int WN = EmWin<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsem[];
    unsigned my_spill = 0;
    T v[6][PPT];
    // the lane's first particles of an item: asked for BEFORE any window work, so that the trips overlap
    auto first_group = [&](const BlockWork& w, size_t& g, size_t& g_end) {
        size_t g_begin;
        fesgrp::groups_exact(w.begin, w.end, PPT, g_begin, g_end); // (exactly the item's own slots: fes_groups.hpp)
        g = g_begin + threadIdx.x;
        if (g < g_end) {
#pragma unroll
            for (int f = 0; f < 6; ++f) load_lane<T, PPT>(a.slab + f * a.stride, g * PPT, v[f]);
        }
    };
    if constexpr (!FES_EM_PIPE) {
        FPIC_LDS T* lF = (FPIC_LDS T*)ldsem;
        FPIC_LDS unsigned long long* lJ = (FPIC_LDS unsigned long long*)(ldsem + EmWin<T>::kFieldBytes);
        // (the item is read before the count that may send the workgroup home: one memory latency instead of two; the list
        // has a slot for every workgroup of the launch)
        const BlockWork w = t.work[blockIdx.x];
        if (blockIdx.x >= *t.nwork) return;
        if (!in_part(w.tile, t.part, t.tiles_per_layer, t.layer_lo, t.layer_hi)) return;
        int ox, oy, oz;
        em_tile_origin(t, w.tile, ox, oy, oz);
        size_t g, g_end;
        first_group(w, g, g_end);
        em_stage<T, kEmThreads, true>(a, ox, oy, oz, lF, lJ);
        __syncthreads();
        em_push_item<T, kEmThreads>(t, w, ox, oy, oz, lF, lJ, v, g, g_end, my_spill);
        __syncthreads();
        em_flush<T, kEmThreads>(a, ox, oy, oz, lJ);
    } else if constexpr (FES_EM_PIPE == 2) {
        // persistent, the tile's FIELDS double-buffered (the next item's records are staged by the waves before they push this
        // item's particles: no barrier in between), ONE accumulator window: push - barrier - flush - barrier per item
        const uint32_t nwork = *t.nwork, stride = gridDim.x;
        auto next_item = [&](uint32_t idx) {
            while (idx < nwork && !in_part(t.work[idx].tile, t.part, t.tiles_per_layer, t.layer_lo, t.layer_hi)) idx += stride;
            return idx;
        };
        uint32_t cur = next_item(blockIdx.x);
        if (cur >= nwork) return;
        auto fields = [&](int b) { return (FPIC_LDS T*)(ldsem + b * EmWin<T>::kFieldBytes); };
        FPIC_LDS unsigned long long* lJ = (FPIC_LDS unsigned long long*)(ldsem + 2 * EmWin<T>::kFieldBytes);
        BlockWork w = t.work[cur];
        int ox, oy, oz;
        em_tile_origin(t, w.tile, ox, oy, oz);
        size_t g, g_end;
        first_group(w, g, g_end);
        em_stage<T, kEmThreads, true>(a, ox, oy, oz, fields(0), lJ);
        __syncthreads();
        int b = 0;
        for (;;) {
            const uint32_t nxt = next_item(cur + stride);
            BlockWork wn = w;
            int nox = 0, noy = 0, noz = 0;
            if (nxt < nwork) {
                wn = t.work[nxt];
                em_tile_origin(t, wn.tile, nox, noy, noz);
                em_stage<T, kEmThreads, false>(a, nox, noy, noz, fields(b ^ 1), lJ);
            }
            em_push_item<T, kEmThreads>(t, w, ox, oy, oz, fields(b), lJ, v, g, g_end, my_spill);
            if (nxt < nwork) first_group(wn, g, g_end); // (its loads fly across the barriers)
            __syncthreads();
            em_flush<T, kEmThreads>(a, ox, oy, oz, lJ);   // (leaves the accumulators clear)
            if (nxt >= nwork) break;
            __syncthreads();
            cur = nxt; w = wn; ox = nox; oy = noy; oz = noz; b ^= 1;
        }
    } else {
        // persistent: items blockIdx.x, + gridDim.x, ... of the part being launched; every wave of the workgroup walks the
        // same items (the list and the count are the same for all of them), so every barrier is reached by all
        const uint32_t nwork = *t.nwork, stride = gridDim.x;
        auto next_item = [&](uint32_t idx) {
            while (idx < nwork && !in_part(t.work[idx].tile, t.part, t.tiles_per_layer, t.layer_lo, t.layer_hi)) idx += stride;
            return idx;
        };
        uint32_t cur = next_item(blockIdx.x);
        if (cur >= nwork) return;
        auto fields = [&](int b) { return (FPIC_LDS T*)(ldsem + b * EmWin<T>::kBufBytes); };
        auto currents = [&](int b) { return (FPIC_LDS unsigned long long*)(ldsem + b * EmWin<T>::kBufBytes + EmWin<T>::kFieldBytes); };
        BlockWork w = t.work[cur];
        int ox, oy, oz;
        em_tile_origin(t, w.tile, ox, oy, oz);
        size_t g, g_end;
        first_group(w, g, g_end);
        for (int s3 = threadIdx.x; s3 < 3 * WN; s3 += kEmThreads) { currents(0)[s3] = 0ull; currents(1)[s3] = 0ull; }
        em_stage<T, kEmThreads, false>(a, ox, oy, oz, fields(0), currents(0));
        __syncthreads();
        int b = 0, pox = 0, poy = 0, poz = 0;
        bool have_prev = false;
        for (;;) {
            const uint32_t nxt = next_item(cur + stride);
            // the other window: the previous item's currents out, the next item's fields in — no barrier between this and
            // the particles below, which touch this item's window only
            if (have_prev) em_flush<T, kEmThreads>(a, pox, poy, poz, currents(b ^ 1));
            BlockWork wn = w;
            int nox = 0, noy = 0, noz = 0;
            if (nxt < nwork) {
                wn = t.work[nxt];
                em_tile_origin(t, wn.tile, nox, noy, noz);
                em_stage<T, kEmThreads, false>(a, nox, noy, noz, fields(b ^ 1), currents(b ^ 1));
            }
            em_push_item<T, kEmThreads>(t, w, ox, oy, oz, fields(b), currents(b), v, g, g_end, my_spill);
            pox = ox; poy = oy; poz = oz; have_prev = true;
            if (nxt < nwork) first_group(wn, g, g_end); // (its loads fly across the barrier)
            __syncthreads();
            if (nxt >= nwork) break;
            cur = nxt; w = wn; ox = nox; oy = noy; oz = noz; b ^= 1;
        }
        em_flush<T, kEmThreads>(a, pox, poy, poz, currents(b));
    }
    if (my_spill) atomicAdd(t.spilled, static_cast<unsigned long long>(my_spill));
}

// The charge grid of a population binned by the full-EM tiles (density() and the start field of precalc() in that mode):
// the same work list and window as em_push_tiles_kernel with ONE int64 accumulator per node (10.6 KB of LDS, so a CU holds
// many workgroups), eight ds_add_u64 per particle, one flush.  The flat form's eight global atomics per particle are bound
// by the memory side's atomic rate whatever the particle order (125 ms for 5e8 particles, sorted or not): five times a
// step() of the same population, which a host that runs the reference's frame (step + density) would pay every frame.
constexpr int kEmRhoThreads = 256;
template <typename T>
__global__ __launch_bounds__(kEmRhoThreads) void em_rho_tiles_kernel(EmTileArgs<T> t, unsigned long long* __restrict__ rho)
{
    const EmPushArgs<T>& a = t.p;
    constexpr int PPT = Vec16<T>::N;
    constexpr int WX = EmWin<T>::WX, WY = EmWin<T>::WY, WN = EmWin<T>::N;
    __shared__ unsigned long long lrho[WN];
    const BlockWork w = t.work[blockIdx.x];
    if (blockIdx.x >= *t.nwork) return;
    if (!in_part(w.tile, t.part, t.tiles_per_layer, t.layer_lo, t.layer_hi)) return;
    int ox, oy, oz;
    em_tile_origin(t, w.tile, ox, oy, oz);
    for (int s = threadIdx.x; s < WN; s += kEmRhoThreads) lrho[s] = 0ull;
    __syncthreads();
    const GlobalGrid<T> grid{ nullptr, rho, a.nx, a.ny, a.nz, a.held };
    unsigned my_spill = 0;
    size_t g_begin, g_end;
    fesgrp::groups_exact(w.begin, w.end, PPT, g_begin, g_end);
    for (size_t g = g_begin + threadIdx.x; g < g_end; g += kEmRhoThreads) {
        const size_t base = g * PPT;
        T px[PPT], py[PPT], pz[PPT];
        load_lane<T, PPT>(a.slab + 0 * a.stride, base, px);
        load_lane<T, PPT>(a.slab + 1 * a.stride, base, py);
        load_lane<T, PPT>(a.slab + 2 * a.stride, base, pz);
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (!fesgrp::owns(w.begin, w.end, base + q) || px[q] < static_cast<T>(0)) continue; // (x < 0: a migrated slot)
            int i, j, k, wx[2], wy[2], wz[2];
            axis(px[q], a.nx, i, wx[1]); wx[0] = 16384 - wx[1];
            axis(py[q], a.ny, j, wy[1]); wy[0] = 16384 - wy[1];
            axis(pz[q], a.nz, k, wz[1]); wz[0] = 16384 - wz[1];
            const int s0 = em_slot<T>(i, j, k, ox, oy, oz, a.nx, a.ny, a.nz);
            if (s0 < 0) { grid.template deposit<true>(i, j, k, wx, wy, wz, a.Z); ++my_spill; continue; }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        const long long v = weight3(wx[aa], wy[b], wz[c] * a.Z);
                        __hip_atomic_fetch_add(lrho + (s0 + aa + WX * b + WX * WY * c), static_cast<unsigned long long>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < WN; s += kEmRhoThreads) {
        const unsigned long long val = lrho[s];
        if (val == 0ull) continue;
        const int n = s / (WX * WY), rem = s - n * (WX * WY);
        const int m = rem / WX, l = rem - m * WX;
        const int gi = wrap_window(ox + l, a.nx), gj = wrap_window(oy + m, a.ny), gk = wrap_window(oz - a.held.zs0 + n, a.nz);
        if (gk < a.held.nzs) atomicAdd(rho + (static_cast<size_t>(gi) + static_cast<size_t>(a.nx) * (static_cast<size_t>(gj) + static_cast<size_t>(a.ny) * gk)), val);
    }
    if (my_spill) atomicAdd(t.spilled, static_cast<unsigned long long>(my_spill));
}

// ------------------------------------------------------------------ spatial decomposition (z-slabs), SURVEY.md 8(e) row 2

// One migrating particle on the wire: the six coordinates and the caller's (global) index.
template <typename T>
struct MigRecord {
    T v[6];
    uint32_t id;
    uint32_t slot;   // where the sender held it (keeps the record at 8-byte multiples; rounds 2-3 put a called-off migration back through it)
};

// Particles whose cell has left this rank's slab [z0, z0 + nzl) are appended to the send buffer of the
// neighbour that owns it (0: the slab below, 1: the slab above) and their slot is marked dead (x = -1);
// the re-binning that follows drops dead slots.  A particle further than `reach` planes from the slab has
// outrun the ghost planes (its charge was lost from the exchange): counted in lost.
constexpr int kMigPer = 16; // particles per lane of the pack: one reservation on the message counters per 4096 particles
// COUNT_ONLY: the same scan, the counters only — nothing is packed and no slot is touched (the count of every species
// is agreed by all ranks before the first particle of any species moves: fes_api.hip, migrate).
template <typename T, bool COUNT_ONLY = false>
__global__ __launch_bounds__(256) void mig_pack_kernel(T* slab, size_t stride, const uint32_t* __restrict__ id, size_t n, int nz, int z0, int nzl,
                                                       int reach, int world, MigRecord<T>* down, MigRecord<T>* up, unsigned cap,
                                                       unsigned* __restrict__ counts /* down, up, lost, overflow */,
                                                       uint32_t* __restrict__ census = nullptr, int nx = 0, int ny = 0, int ntx = 0, int nty = 0,
                                                       const uint32_t* __restrict__ tile_start = nullptr, uint32_t t_lo = 0, uint32_t t_hi = 0)
{
    // (tile_start given: the slots of the tiles [t_lo, t_hi) — the interior layers of the slab, at least one tile layer
    // from either face — are skipped: nobody can have left the slab from there since the last binning)
    const size_t skip_from = tile_start ? tile_start[t_lo] : 0, skip = tile_start ? tile_start[t_hi] - tile_start[t_lo] : 0;
    // (the leavers of a sorted array sit in the tiles along the two faces: counted in LDS first, so that the two
    // message counters see one atomic per workgroup and direction instead of one per particle)
    __shared__ unsigned l_cnt[2], l_base[2], l_lost;
    __shared__ TileTally tally;
    // a fixed grid walks the chunks of the slots that are scanned (n - skip of them): with three quarters of a slab's
    // slots skipped, a grid sized for n spent most of the pass launching workgroups that had nothing to do
    const size_t live = n - skip;
    for (size_t first = static_cast<size_t>(blockIdx.x) * (256 * kMigPer); first < live; first += static_cast<size_t>(gridDim.x) * (256 * kMigPer)) {
        if (threadIdx.x < 2) l_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 2) l_lost = 0;
        tally_reset(tally);
        uint32_t flags = 0; // two bits per particle of this lane: 1 = down, 2 = up
        T xs[kMigPer], zs[kMigPer];
#pragma unroll
        for (int k = 0; k < kMigPer; ++k) { // (all loads first: one latency for the sixteen slots, not sixteen)
            size_t s = first + static_cast<size_t>(k) * 256 + threadIdx.x;
            if (s >= skip_from) s += skip;
            const bool in = s < n;
            xs[k] = in ? slab[s] : static_cast<T>(-1);
            zs[k] = in ? slab[2 * stride + s] : static_cast<T>(0);
        }
#pragma unroll
        for (int k = 0; k < kMigPer; ++k) {
            if (xs[k] < static_cast<T>(0)) continue; // (beyond the array, or the slot of a particle that has left)
            int kz, w;
            axis(zs[k], nz, kz, w);
            int d = kz - z0;                    // planes above the slab's first, periodic
            if (d < 0) d += nz;
            if (d < nzl) continue;              // still at home
            const int above = d - nzl, below = nz - d - 1; // cells beyond the upper / lower face
            const bool go_up = world == 2 ? true : above <= below;
            if ((above <= below ? above : below) >= reach) atomicAdd(&l_lost, 1u);
            atomicAdd(&l_cnt[go_up ? 1 : 0], 1u);
            flags |= (go_up ? 2u : 1u) << (2 * k);
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const unsigned c = l_cnt[threadIdx.x];
            l_base[threadIdx.x] = c ? atomicAdd(counts + threadIdx.x, c) : 0u;
            l_cnt[threadIdx.x] = 0;
        }
        if (threadIdx.x == 2 && l_lost) atomicAdd(counts + 2, l_lost);
        __syncthreads();
        if constexpr (COUNT_ONLY) continue; // (every thread of the workgroup: the barriers above are the round's last)
#pragma unroll
        for (int k = 0; k < kMigPer; ++k) {
            const unsigned dir = (flags >> (2 * k)) & 3u;
            if (!dir) continue;
            size_t s = first + static_cast<size_t>(k) * 256 + threadIdx.x;
            if (s >= skip_from) s += skip;
            const unsigned slot = l_base[dir - 1] + atomicAdd(&l_cnt[dir - 1], 1u);
            if (slot >= cap) { atomicAdd(counts + 3, 1u); continue; } // stays (and is reported): no room in the message
            MigRecord<T> r;
#pragma unroll
            for (int f = 0; f < 6; ++f) r.v[f] = slab[f * stride + s];
            r.id = id[s];
            r.slot = static_cast<uint32_t>(s);
            (dir == 2 ? up : down)[slot] = r;
            slab[s] = static_cast<T>(-1);
            // the census of the last push counted it in its tile: the next bin table is laid out without it
            if (census) tally_add<false>(tally, key_of<T>(r.v[0], r.v[1], r.v[2], nx, ny, nz, ntx, nty), census);
        }
        if (census) tally_flush<false>(tally, census);
        __syncthreads(); // (the counters are reset at the top of the next round)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mig_append_kernel(const MigRecord<T>* __restrict__ in, unsigned count, T* slab, size_t stride, uint32_t* id, size_t first,
                                                         uint32_t* __restrict__ census = nullptr, int nx = 0, int ny = 0, int nz = 0, int ntx = 0, int nty = 0)
{
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < count;
    MigRecord<T> m{};
    if (active) {
        m = in[r];
#pragma unroll
        for (int f = 0; f < 6; ++f) slab[f * stride + first + r] = m.v[f];
        id[first + r] = m.id;
    }
    if (census) { // ... and with the arrivals: one atomic per run of records of one tile
        const uint32_t key = active ? key_of<T>(m.v[0], m.v[1], m.v[2], nx, ny, nz, ntx, nty) : ~0u;
        int head, len;
        wave_runs(key, head, len);
        if (active && head == static_cast<int>(threadIdx.x & 63)) atomicAdd(census + key, static_cast<uint32_t>(len));
    }
}

// the two message counters counted every leaver; the messages hold at most `cap` records each (the others stayed)
// (record_overflow, after a COUNT_ONLY scan: counts[3] <- the leavers that will stay behind, which the pack itself counts)
static __global__ void mig_clamp_kernel(unsigned* counts, unsigned cap, int record_overflow = 0)
{
    if (threadIdx.x < 2 && counts[threadIdx.x] > cap) {
        if (record_overflow) atomicAdd(counts + 3, counts[threadIdx.x] - cap);
        counts[threadIdx.x] = cap;
    }
}

// ghost planes received from a neighbour, added onto this rank's own planes (exact: int64)
static __global__ __launch_bounds__(256) void ghost_add_kernel(long long* __restrict__ dst, const long long* __restrict__ src, size_t count)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < count) dst[c] += src[c];
}

// ---- slab-decomposed Poisson solve: the half spectrum of the owned planes, hatA [nzl][ny][nxh] (complex T), is
// transposed over the ranks into hatB [nz][nyl][nxh] (all z, the rank's share of the ky rows) and back

// hatA -> send buffer [q][nzl][nyl][nxh]: the block of rows that rank q transforms along z
template <typename T>
__global__ __launch_bounds__(256) void transpose_pack_kernel(const T* __restrict__ hatA, int nxh, int ny, int nzl, int nyl, T* __restrict__ send)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t total = static_cast<size_t>(nxh) * ny * nzl;
    if (c >= total) return;
    const int x = static_cast<int>(c % nxh), y = static_cast<int>((c / nxh) % ny), z = static_cast<int>(c / (static_cast<size_t>(nxh) * ny));
    const int q = y / nyl, yl = y - q * nyl;
    const size_t d = ((static_cast<size_t>(q) * nzl + z) * nyl + yl) * nxh + x;
    send[2 * d] = hatA[2 * c];
    send[2 * d + 1] = hatA[2 * c + 1];
}

// receive buffer [q][nzl][nyl][nxh] (what rank q transformed of MY planes) -> hatA
template <typename T>
__global__ __launch_bounds__(256) void transpose_unpack_kernel(const T* __restrict__ recv, int nxh, int ny, int nzl, int nyl, T* __restrict__ hatA)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t total = static_cast<size_t>(nxh) * ny * nzl;
    if (c >= total) return;
    const int x = static_cast<int>(c % nxh), y = static_cast<int>((c / nxh) % ny), z = static_cast<int>(c / (static_cast<size_t>(nxh) * ny));
    const int q = y / nyl, yl = y - q * nyl;
    const size_t d = ((static_cast<size_t>(q) * nzl + z) * nyl + yl) * nxh + x;
    hatA[2 * c] = recv[2 * d];
    hatA[2 * c + 1] = recv[2 * d + 1];
}

// [rows][cols] -> [cols][rows] of complex values through a 32 x 32 LDS tile (both sides coalesced).  The transposition
// lands as hatB [nz][nyl][nxh]: a transform along z there has a stride of nyl * nxh, which rocFFT runs at 0.6 TB/s
// (len512 "sbrr", 0.23 ms for a rank's 8.4e6 points); turned to [nyl * nxh][nz] the same transform is contiguous.
template <typename T>
__global__ __launch_bounds__(256) void transpose_complex_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols)
{
    __shared__ T tile[32][33][2];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8 threads
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) {
            const size_t s = 2 * (static_cast<size_t>(r) * cols + c);
            tile[k][tx][0] = in[s]; tile[k][tx][1] = in[s + 1];
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx; // out[c][r]
        if (r < rows && c < cols) {
            const size_t d = 2 * (static_cast<size_t>(c) * rows + r);
            out[d] = tile[tx][k][0]; out[d + 1] = tile[tx][k][1];
        }
    }
}

// phi_hat = rho_hat / (eps0 K^2 N) on hatZ [nyl][nxh][nz] (z contiguous): ky = y0 + yl
template <typename T>
__global__ __launch_bounds__(256) void kspace_zmajor_kernel(T* __restrict__ hatZ, int nxh, int nyl, int nz, int y0, const double* __restrict__ k2x,
                                                            const double* __restrict__ k2y, const double* __restrict__ k2z, double inv_eps0_n)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nxh) * nyl * nz) return;
    const int k = static_cast<int>(c % nz);
    const size_t m = c / nz;
    const int i = static_cast<int>(m % nxh), j = y0 + static_cast<int>(m / nxh);
    const double K2 = (k2x[i] + k2y[j]) + k2z[k];
    const T g = (i | j | k) ? static_cast<T>(inv_eps0_n / K2) : static_cast<T>(0);
    hatZ[2 * c] = hatZ[2 * c] * g;
    hatZ[2 * c + 1] = hatZ[2 * c + 1] * g;
}

// phi_hat = rho_hat / (eps0 K^2 N) on hatB [nz][nyl][nxh]: ky = y0 + yl (kspace_kernel of the decomposed solve)
template <typename T>
__global__ __launch_bounds__(256) void kspace_slab_kernel(T* __restrict__ hatB, int nxh, int nyl, int nz, int y0, const double* __restrict__ k2x,
                                                          const double* __restrict__ k2y, const double* __restrict__ k2z, double inv_eps0_n)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nxh) * nyl * nz) return;
    const int i = static_cast<int>(c % nxh), j = y0 + static_cast<int>((c / nxh) % nyl), k = static_cast<int>(c / (static_cast<size_t>(nxh) * nyl));
    const double K2 = (k2x[i] + k2y[j]) + k2z[k];
    const T g = (i | j | k) ? static_cast<T>(inv_eps0_n / K2) : static_cast<T>(0);
    hatB[2 * c] = hatB[2 * c] * g;
    hatB[2 * c + 1] = hatB[2 * c + 1] * g;
}

// gradient_kernel on the planes k0 .. k0 + count - 1 (periodic); phi and E4 hold the planes `held` (the planes named and
// one beyond on either side are held)
template <typename T>
__global__ __launch_bounds__(256) void gradient_planes_kernel(const T* __restrict__ phi, int nx, int ny, int nz, int k0, int count, T hx, T hy, T hz,
                                                              T* __restrict__ E4, Held held)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x), j = static_cast<int>(blockIdx.y * blockDim.y + threadIdx.y); // (node_launch)
    if (i >= nx || j >= ny || static_cast<int>(blockIdx.z) >= count) return;
    const size_t sy = static_cast<size_t>(nx), sz = static_cast<size_t>(nx) * ny;
    int k = (k0 + static_cast<int>(blockIdx.z)) % nz;   // (uniform over the workgroup: scalar arithmetic)
    if (k < 0) k += nz;
    const int im = i ? i - 1 : nx - 1, ip = (i + 1 == nx) ? 0 : i + 1;
    const int jm = j ? j - 1 : ny - 1, jp = (j + 1 == ny) ? 0 : j + 1;
    const int km = held_plane(k ? k - 1 : nz - 1, held, nz), kp = held_plane((k + 1 == nz) ? 0 : k + 1, held, nz);
    k = held_plane(k, held, nz);
    if ((k | km | kp) < 0) return; // (a plane the caller should not have named)
    const size_t c = i + sy * j + sz * k;
    using V = typename NatVec16<T>::type;
    const T o0 = (phi[im + sy * j + sz * k] - phi[ip + sy * j + sz * k]) * hx;
    const T o1 = (phi[i + sy * jm + sz * k] - phi[i + sy * jp + sz * k]) * hy;
    const T o2 = (phi[i + sy * j + sz * km] - phi[i + sy * j + sz * kp]) * hz;
    const T o3 = phi[c];
    if constexpr (sizeof(T) == 4) {
        V v; v.x = o0; v.y = o1; v.z = o2; v.w = o3;
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(E4 + 4 * c)); // (as gradient_kernel)
    } else {
        V v0, v1; v0.x = o0; v0.y = o1; v1.x = o2; v1.y = o3;
        __builtin_nontemporal_store(v0, reinterpret_cast<V*>(E4 + 4 * c));
        __builtin_nontemporal_store(v1, reinterpret_cast<V*>(E4 + 4 * c + 2));
    }
}

// ids of a freshly uploaded population: first + slot
// checkpoint: the raw state of the caller's particles [first, first + m) <-> an AoS staging buffer (6 T per particle)
template <typename T>
__global__ __launch_bounds__(256) void ckpt_gather_kernel(const T* __restrict__ slab, size_t stride, const uint32_t* __restrict__ id, size_t n, size_t first, size_t m,
                                                          T* __restrict__ stage)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = static_cast<size_t>(id[s]) - first; // (unsigned: below `first` wraps past m)
    if (i >= m) return;
#pragma unroll
    for (int f = 0; f < 6; ++f) stage[6 * i + f] = slab[f * stride + s];
}

template <typename T>
__global__ __launch_bounds__(256) void ckpt_scatter_kernel(T* __restrict__ slab, size_t stride, uint32_t* __restrict__ id, size_t first, size_t m,
                                                           const T* __restrict__ stage)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= m) return;
#pragma unroll
    for (int f = 0; f < 6; ++f) slab[f * stride + first + i] = stage[6 * i + f];
    id[first + i] = static_cast<uint32_t>(first + i);
}

static __global__ __launch_bounds__(256) void iota3_kernel(uint32_t* id, size_t n, uint32_t first)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) id[i] = first + static_cast<uint32_t>(i);
}

template <typename T, typename Out>
__global__ __launch_bounds__(256) void get_plain3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, size_t n, Out* __restrict__ aos)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    aos[3 * s] = static_cast<Out>(a[s]); aos[3 * s + 1] = static_cast<Out>(b[s]); aos[3 * s + 2] = static_cast<Out>(c[s]);
}

} // namespace fes
