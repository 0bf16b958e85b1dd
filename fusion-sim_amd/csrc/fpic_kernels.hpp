// fpic_kernels.hpp — CDNA4 (gfx950) kernels of the particle-in-cell hot path.
//
// One template per kernel, instantiated for float (the reference's precision) and
// double.  Each kernel names the reference GLSL program it computes
// (/root/reference/public/javascripts/empic.js).  Expression order follows the
// shader text; the library is built with -ffp-contract=off and correctly rounded
// sqrt / divide so that float results agree with the CPU oracle bit for bit.
//
// Device layout
//   particles : structure of arrays x,y,z,vx,vy,vz,u1,u2,c1,c2 (T) + alive (u8) + id (u32);
//               16-byte vector loads, 4 (float) or 2 (double) particles per lane.
//               id is the caller's particle index: binning permutes the arrays.
//   coef      : 12 T per cell, [R11 R12 R13 A1 | R21 R22 R23 A2 | R31 R32 R33 A3],
//               cell = i + j*nr, so a lane's gather is three 16-byte loads of one record.
//   sink_alive: 1 byte per cell (sink.r > 0.5 evaluated once at set()).
//   inv_cdf_xy: 2 T per texel of the 512x512 injection table.
//   entropy   : 4 T per texel of the 1024x1024 table.
//   cell_sums : 4 T per cell of an (nr+1) x (nz+1) grid plus an apron of kSumsApron cells (fpic_internal.hpp):
//               sums of 0.001*(vr,vtheta,vz,1) over the particles whose sprite is centred on the cell.
#pragma once

#include <hip/hip_runtime.h>

#include "fpic_internal.hpp"

namespace fpic {

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ T sqrt_(T v);
template <> __device__ __forceinline__ float sqrt_<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double sqrt_<double>(double v) { return sqrt(v); }
template <typename T> __device__ __forceinline__ T cos_(T v);
template <> __device__ __forceinline__ float cos_<float>(float v) { return cosf(v); }
template <> __device__ __forceinline__ double cos_<double>(double v) { return cos(v); }

template <typename T>
struct ParticleArrays {
    T *x, *y, *z, *vx, *vy, *vz, *u1, *u2, *c1, *c2;
    uint8_t* alive;
    uint32_t* id;
};

struct BlockWork {
    uint32_t tile, begin, end, pad;
};

// NEAREST + CLAMP_TO_EDGE texel index (utilities.js:528-531); NaN selects texel 0.
template <typename T>
__device__ __forceinline__ int ngp(T u, int W)
{
    // branch-free: fmax drops a NaN (-> 0), the upper clamp at W-1 equals floor's result
    // for every t in [W-1, W) and clamps everything above (W-1 is exact in T for W <= 2^24)
    const T t = u * static_cast<T>(W);
    if constexpr (sizeof(T) == 4) return static_cast<int>(fminf(fmaxf(t, 0.0f), static_cast<float>(W - 1)));
    else return static_cast<int>(fmin(fmax(t, 0.0), static_cast<double>(W - 1)));
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, T (&o)[4])
{
    if constexpr (sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        const double2 a = *reinterpret_cast<const double2*>(p);
        const double2 b = *reinterpret_cast<const double2*>(p + 2);
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    }
}

// Particle streams are read once and written once per launch: they are accessed with
// the non-temporal hint so that they do not push the gathered tables (entropy,
// coefficients) out of the XCD's L2.
typedef float nat_f32x4 __attribute__((ext_vector_type(4)));
typedef double nat_f64x2 __attribute__((ext_vector_type(2)));
template <typename T> struct NatVec16;
template <> struct NatVec16<float> { using type = nat_f32x4; };
template <> struct NatVec16<double> { using type = nat_f64x2; };

// (Tried and rejected on gfx950, profiles/r01_cache_policy.txt: sc1 / sc0 sc1 write-through
// stores and sc1 loads through buffer instructions; none beat nt loads + nt stores.)
template <typename T, int N>
__device__ __forceinline__ void load_lane(const T* arr, size_t base, T (&o)[N])
{
    using V = typename NatVec16<T>::type;
    static_assert(N == Vec16<T>::N, "one 16-byte vector per lane");
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(arr + base));
    if constexpr (N == 4) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else { o[0] = v.x; o[1] = v.y; }
}

template <typename T, int N>
__device__ __forceinline__ void store_lane(T* arr, size_t base, const T (&o)[N])
{
    using V = typename NatVec16<T>::type;
    V v;
    if constexpr (N == 4) { v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3]; }
    else { v.x = o[0]; v.y = o[1]; }
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(arr + base));
}

// ------------------------------------------------------------------ scatter (K4), stage 1
//
// programMoments01 (empic.js:980-1035) draws every particle as an 11x11 sprite whose
// texel weights do not depend on the sub-cell offset (NEAREST lookup of
// gl_PointCoord).  The blended result is therefore the per-cell sums of the vertex
// colour 0.001*(vr,vtheta,vz,1) convolved with the stamp.  Stage 1 forms the
// per-cell sums; stage 2 (stamp_finish_kernel) applies the stamp.

// Deposit cell of a particle, or false when the point is clipped (centre outside
// the clip volume, or NaN).  ic in [0,nr], jc in [0,nz]: r = 1 lands on column nr.
template <typename T>
__device__ __forceinline__ bool deposit_cell(T x, T y, T z, int nr, int nz, T& r, int& ic, int& jc)
{
    r = sqrt_(x * x + y * y);
    if (!(r >= static_cast<T>(0) && r <= static_cast<T>(1) && z >= static_cast<T>(0) && z <= static_cast<T>(1)))
        return false;
    ic = static_cast<int>(r * static_cast<T>(nr));
    jc = static_cast<int>(z * static_cast<T>(nz));
    return true;
}

// The same under the rasterised convention (spec.raster_subpixel_bits = b > 0; the CPU restatement is deposit_raster,
// pinned by tests/golden/webgl_*): what a rasteriser with b sub-pixel bits makes of the
// point.  Clip coordinate 2u-1, viewport transform to a fixed-point window position in pixel-centre coordinates with
// y running downwards, X = rint(X0 + ndc*Wb), one rounding per operation; the sprite is the square X +- 11*2^(b-1),
// left/top edges inclusive; its first column (row from the top) is ceil((X - 11*2^(b-1)) / 2^b).  Returns the
// first column (y_down = false) or the first row counted from the bottom (y_down = true), i.e. centre - 5; false
// for a NaN / infinite / absurd coordinate.  The footprint may leave the grid: it is cropped, not dropped.
template <typename T>
__device__ __forceinline__ bool raster_first(T u, int W, int bits, bool y_down, int& first)
{
    const T ndc = static_cast<T>(2) * u - static_cast<T>(1);
    const T wb = static_cast<T>(W) * static_cast<T>(0.5) * static_cast<T>(1 << bits);
    const T x0 = wb - static_cast<T>(1 << bits) * static_cast<T>(0.5);
    const T t = ndc * (y_down ? -wb : wb);
    const T s = x0 + t;
    if (!(s > static_cast<T>(-1073741824.0) && s < static_cast<T>(1073741824.0))) return false;
    int X;
    if constexpr (sizeof(T) == 4) X = __float2int_rn(s); else X = __double2int_rn(s);   // round half to even
    const int a = X - 11 * (1 << bits) / 2;
    const int p0 = (a + (1 << bits) - 1) >> bits;                                        // ceil(a / 2^b)
    first = y_down ? W - 1 - (p0 + 10) : p0;
    return true;
}

// Centre cell of the rasterised sprite of (r, z); false when nothing of it can reach the grid.
template <typename T>
__device__ __forceinline__ bool raster_cell(T r, T z, int nr, int nz, int bits, int& ic, int& jc)
{
    int i0, j0;
    if (!raster_first(r, nr, bits, false, i0) || !raster_first(z, nz, bits, true, j0)) return false;
    if (i0 >= nr || i0 + 10 < 0 || j0 >= nz || j0 + 10 < 0) return false;
    ic = i0 + 5; jc = j0 + 5;
    return true;
}

// One workgroup sums one chunk of the particles binned to one tile.  The tile and an
// 8-cell halo live in LDS as DOUBLE accumulators: on gfx950 ds_add_f32 sustains only
// ~0.4 lanes/clk/CU while ds_add_f64 is 3.7x and ds_add_u32 9x faster (measured,
// profiles/r01_lds_atomics.txt), and double sums are also closer to the exact value
// than any float ordering.  The non-zero part of the tile is then flushed with global
// float atomics in 256-byte contiguous pieces.  A particle that has drifted beyond the
// halo since the last binning goes straight to global memory (counted in *spilled);
// correctness never depends on the binning.
constexpr int kSumsThreads = 512;
constexpr size_t kSumsLdsBytes = static_cast<size_t>(kTileLds) * kTileLds * 4 * sizeof(double);

template <typename T>
__global__ __launch_bounds__(kSumsThreads) void cell_sums_kernel(ParticleArrays<T> p, int nr, int nz, int ntx,
                                                                 const BlockWork* __restrict__ work,
                                                                 const uint32_t* __restrict__ nwork,
                                                                 T* __restrict__ cell_sums, unsigned long long* spilled,
                                                                 int raster_bits)
{
    constexpr int PPT = Vec16<T>::N;
    constexpr int LW = kTileLds;
    constexpr int BS = kSumsThreads;
    extern __shared__ double tile[];
    if (blockIdx.x >= *nwork) return;
    const BlockWork w = work[blockIdx.x];
    const int i0 = static_cast<int>(w.tile % ntx) * kTileSide - kTileHalo;
    const int j0 = static_cast<int>(w.tile / ntx) * kTileSide - kTileHalo;
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) tile[k] = 0.0;
    __syncthreads();

    unsigned int my_spill = 0;
    const size_t first = (static_cast<size_t>(w.begin) / PPT) * PPT;
    for (size_t base = first + static_cast<size_t>(threadIdx.x) * PPT; base < w.end; base += BS * PPT) {
        T x[PPT], y[PPT], z[PPT], vx[PPT], vy[PPT], vz[PPT];
        load_lane<T, PPT>(p.x, base, x);
        load_lane<T, PPT>(p.y, base, y);
        load_lane<T, PPT>(p.z, base, z);
        load_lane<T, PPT>(p.vx, base, vx);
        load_lane<T, PPT>(p.vy, base, vy);
        load_lane<T, PPT>(p.vz, base, vz);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const size_t i = base + k;
            if (i < w.begin || i >= w.end) continue;
            T r;
            int ic, jc;
            const bool inside = deposit_cell(x[k], y[k], z[k], nr, nz, r, ic, jc);
            if (raster_bits ? !raster_cell(r, z[k], nr, nz, raster_bits, ic, jc) : !inside) continue;
            const T dx = x[k] / r, dy = y[k] / r;
            const T vr = vx[k] * dx + vy[k] * dy;
            const T va = vy[k] * dx - vx[k] * dy;
            // v_color of the vertex shader, in T as the reference computes it (empic.js:1006)
            const T c0 = static_cast<T>(0.001) * vr;
            const T c1 = static_cast<T>(0.001) * va;
            const T c2 = static_cast<T>(0.001) * vz[k];
            const T c3 = static_cast<T>(0.001) * static_cast<T>(1);
            const int li = ic - i0, lj = jc - j0;
            if (li >= 0 && li < LW && lj >= 0 && lj < LW) {
                double* t = tile + 4 * (lj * LW + li);
                atomicAdd(t, static_cast<double>(c0));
                atomicAdd(t + 1, static_cast<double>(c1));
                atomicAdd(t + 2, static_cast<double>(c2));
                atomicAdd(t + 3, static_cast<double>(c3));
            } else {
                T* g = cell_sums + 4 * sums_index(ic, jc, nr);
                atomicAdd(g, c0);
                atomicAdd(g + 1, c1);
                atomicAdd(g + 2, c2);
                atomicAdd(g + 3, c3);
                if (inside) ++my_spill;     // a point outside the unit square is not a sign of stale bins
            }
        }
    }
    __syncthreads();
    // flush: consecutive lanes take consecutive scalars of one LDS row = consecutive
    // global addresses, so a wave's atomic is one 256-byte piece
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) {
        const double v = tile[k];
        if (v == 0.0) continue;
        const int lj = k / (LW * 4);
        const int rem = k - lj * (LW * 4);
        const int gi = i0 + (rem >> 2), gj = j0 + lj;
        if (!sums_holds(gi, gj, nr, nz)) continue;
        atomicAdd(cell_sums + 4 * sums_index(gi, gj, nr) + (rem & 3), static_cast<T>(v));
    }
    if (my_spill) atomicAdd(spilled, static_cast<unsigned long long>(my_spill));
}

// EXTENSION spec.shape = 1 (SURVEY.md 8(b) key shape:'cic'; no reference counterpart): the vertex colour spread
// bilinearly over the four cell centres around the point (oracle: orc_*_deposit_cic).  Same workgroup, window
// and flush as cell_sums_kernel; the per-cell grid then IS the moments grid (the finish stage runs with a
// one-cell stamp).  16 LDS atomics per particle against 4 for the reference's shape.
template <typename T>
__global__ __launch_bounds__(kSumsThreads) void cic_sums_kernel(ParticleArrays<T> p, int nr, int nz, int ntx,
                                                                const BlockWork* __restrict__ work,
                                                                const uint32_t* __restrict__ nwork,
                                                                T* __restrict__ cell_sums, unsigned long long* spilled)
{
    constexpr int PPT = Vec16<T>::N;
    constexpr int LW = kTileLds;
    constexpr int BS = kSumsThreads;
    extern __shared__ double tile[];
    if (blockIdx.x >= *nwork) return;
    const BlockWork w = work[blockIdx.x];
    const int i0 = static_cast<int>(w.tile % ntx) * kTileSide - kTileHalo;
    const int j0 = static_cast<int>(w.tile / ntx) * kTileSide - kTileHalo;
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) tile[k] = 0.0;
    __syncthreads();
    unsigned int my_spill = 0;
    const size_t first = (static_cast<size_t>(w.begin) / PPT) * PPT;
    for (size_t base = first + static_cast<size_t>(threadIdx.x) * PPT; base < w.end; base += BS * PPT) {
        T x[PPT], y[PPT], z[PPT], vx[PPT], vy[PPT], vz[PPT];
        load_lane<T, PPT>(p.x, base, x);
        load_lane<T, PPT>(p.y, base, y);
        load_lane<T, PPT>(p.z, base, z);
        load_lane<T, PPT>(p.vx, base, vx);
        load_lane<T, PPT>(p.vy, base, vy);
        load_lane<T, PPT>(p.vz, base, vz);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const size_t i = base + k;
            if (i < w.begin || i >= w.end) continue;
            T r;
            int ic, jc;
            if (!deposit_cell(x[k], y[k], z[k], nr, nz, r, ic, jc)) continue;
            const T dx = x[k] / r, dy = y[k] / r;
            const T col[4] = { static_cast<T>(0.001) * (vx[k] * dx + vy[k] * dy), static_cast<T>(0.001) * (vy[k] * dx - vx[k] * dy),
                               static_cast<T>(0.001) * vz[k], static_cast<T>(0.001) * static_cast<T>(1) };
            const T gi = r * static_cast<T>(nr) - static_cast<T>(0.5), gj = z[k] * static_cast<T>(nz) - static_cast<T>(0.5);
            const T fi0 = sizeof(T) == 4 ? floorf(gi) : floor(gi), fj0 = sizeof(T) == 4 ? floorf(gj) : floor(gj);
            const int ci0 = static_cast<int>(fi0), cj0 = static_cast<int>(fj0);
            T wr[2], wz[2];
            wr[1] = gi - fi0; wr[0] = static_cast<T>(1) - wr[1];
            wz[1] = gj - fj0; wz[0] = static_cast<T>(1) - wz[1];
            bool spilled_here = false;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int j = cj0 + b;
                if (j < 0 || j >= nz) continue;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int ii = ci0 + a;
                    if (ii < 0 || ii >= nr) continue;
                    const T wgt = wr[a] * wz[b];
                    const int li = ii - i0, lj = j - j0;
                    if (li >= 0 && li < LW && lj >= 0 && lj < LW) {
                        double* t = tile + 4 * (lj * LW + li);
#pragma unroll
                        for (int c = 0; c < 4; ++c) atomicAdd(t + c, static_cast<double>(col[c] * wgt));
                    } else {
                        T* g = cell_sums + 4 * sums_index(ii, j, nr);
#pragma unroll
                        for (int c = 0; c < 4; ++c) atomicAdd(g + c, col[c] * wgt);
                        spilled_here = true;
                    }
                }
            }
            if (spilled_here) ++my_spill;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) {
        const double v = tile[k];
        if (v == 0.0) continue;
        const int lj = k / (LW * 4);
        const int rem = k - lj * (LW * 4);
        const int gi = i0 + (rem >> 2), gj = j0 + lj;
        if (!sums_holds(gi, gj, nr, nz)) continue;
        atomicAdd(cell_sums + 4 * sums_index(gi, gj, nr) + (rem & 3), static_cast<T>(v));
    }
    if (my_spill) atomicAdd(spilled, static_cast<unsigned long long>(my_spill));
}

// ------------------------------------------------------------------ scatter stage 2 + K5 + K6 + K7
//
// moments01 = stamp (*) cell_sums, cropped at the grid's edges (empic.js:1473-1478);
// programNormalizeMoments01 (empic.js:1052-1057); avg_frag with u_ratio
// (empic.js:273-278, :1083); the avgA -> avgB copy (empic.js:1490-1495) is the
// in-place update of the single avg buffer.
// A particle in cell (ic,jc) adds weight[(di+5) + 11*(5-dj)] to cell (ic+di, jc+dj)
// (gl_PointCoord.t runs downwards; the stamp is symmetric).
template <typename T>
__global__ __launch_bounds__(256) void stamp_finish_kernel(const T* __restrict__ cell_sums, int nr, int nz,
                                                           const float* __restrict__ stamp, T* __restrict__ moments,
                                                           T* __restrict__ norm, T* __restrict__ avg, T ratio, int identity = 0)
{
    constexpr int OT = 32;                      // output tile edge
    constexpr int LW = OT + 2 * kStampReach;    // staged cell_sums tile edge
    __shared__ T g[LW * LW * 4];
    __shared__ float w[kStampCells];
    const int i0 = blockIdx.x * OT, j0 = blockIdx.y * OT;
    for (int k = threadIdx.x; k < kStampCells; k += 256) w[k] = stamp[k];
    for (int k = threadIdx.x; k < LW * LW; k += 256) {
        const int lj = k / LW, li = k - lj * LW;
        const int gi = i0 - kStampReach + li, gj = j0 - kStampReach + lj;
        T v[4] = { 0, 0, 0, 0 };
        if (sums_holds(gi, gj, nr, nz)) load4(cell_sums + 4 * sums_index(gi, gj, nr), v);
        g[4 * k] = v[0]; g[4 * k + 1] = v[1]; g[4 * k + 2] = v[2]; g[4 * k + 3] = v[3];
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const T keep = static_cast<T>(1) - ratio;
    for (int m = 0; m < OT / 8; ++m) {
        const int lj = ty + 8 * m;
        const int i = i0 + tx, j = j0 + lj;
        if (i >= nr || j >= nz) continue;
        T acc[4] = { 0, 0, 0, 0 };
        if (identity) { // shape 'cic': the per-cell grid is the moments grid (and 0 x NaN of a neighbour must not leak in)
            const T* s = g + 4 * ((lj + kStampReach) * LW + (tx + kStampReach));
            acc[0] = s[0]; acc[1] = s[1]; acc[2] = s[2]; acc[3] = s[3];
        }
        for (int b = 0; b < (identity ? 0 : kStampSide); ++b) {
#pragma unroll
            for (int a = 0; a < kStampSide; ++a) {
                // source cell (i - di, j - dj) with di = 5 - a, dj = 5 - b
                const T wt = static_cast<T>(w[(10 - a) + kStampSide * b]);
                const T* s = g + 4 * ((lj + b) * LW + (tx + a));
                acc[0] += wt * s[0];
                acc[1] += wt * s[1];
                acc[2] += wt * s[2];
                acc[3] += wt * s[3];
            }
        }
        const size_t c = 4 * (static_cast<size_t>(i) + static_cast<size_t>(nr) * j);
        const T xc = (static_cast<T>(i) + static_cast<T>(0.5)) / static_cast<T>(nr);
        T nm[4] = { 0, 0, 0, 0 };
        if (acc[3] > static_cast<T>(0)) {
            nm[0] = acc[0] / acc[3]; nm[1] = acc[1] / acc[3]; nm[2] = acc[2] / acc[3]; nm[3] = acc[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T nv = static_cast<T>(1000) * nm[k] * static_cast<T>(0.5) / xc;
            moments[c + k] = acc[k];
            norm[c + k] = nv;
            avg[c + k] = ratio * nv + keep * avg[c + k];
        }
    }
}

// ------------------------------------------------------------------ precalc (K8 + K9)

// programPre1/2/3 and programPreA (empic.js:506-659) in one pass over the cells.
// Bf/Ef are RGBA; h is u_h; f_rz, f_zr, fr, fz are the literals baked into the
// shader text (empic.js:527, :566, :606, :647).
template <typename T>
__global__ __launch_bounds__(256) void precalc_kernel(const T* __restrict__ Bf, const T* __restrict__ Ef, size_t ncell,
                                                      T h, T fr, T fz, T f_rz, T f_zr, int physical_a,
                                                      T* __restrict__ coef)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    T B[4], E[4];
    load4(Bf + 4 * c, B);
    load4(Ef + 4 * c, E);
    const T Bx = B[0], By = B[1], Bz = B[2];
    const T Ex = E[0], Ey = E[1], Ez = E[2];
    const T Bmag = sqrt_((Bx * Bx + By * By) + Bz * Bz);
    const T hB2 = h * h * Bmag * Bmag;
    const T factor = static_cast<T>(2) / (static_cast<T>(1) + hB2);
    const T diag = static_cast<T>(1) - hB2 * factor;
    const T fh = factor * h;
    T* o = coef + 12 * c;

    o[0] = diag + fh * h * Bx * Bx;
    o[1] = fh * (Bz + h * Bx * By);
    o[2] = (fh * (-By + h * Bx * Bz)) * f_rz;

    o[4] = fh * (-Bz + h * By * Bx);
    o[5] = diag + fh * h * By * By;
    o[6] = (fh * (Bx + h * By * Bz)) * f_rz;

    o[8] = (fh * (By + h * Bz * Bx)) * f_zr;
    o[9] = (fh * (-Bx + h * Bz * By)) * f_zr;
    o[10] = diag + fh * h * Bz * Bz;

    const T a = h * (static_cast<T>(2) - hB2 * factor);
    const T b = h * h * factor;
    const T cx = Ey * Bz - Ez * By;
    const T cy = Ez * Bx - Ex * Bz;
    const T cz = Ex * By - Ey * Bx;
    const T dot = (Ex * Bx + Ey * By) + Ez * Bz;
    const T hd = h * dot;
    // quirk Q1 (empic.js:645): the reference adds the scalar u_h*dot(E,B) to every
    // component; physical_a selects h (E.B) B instead
    const T kx = physical_a ? hd * Bx : hd;
    const T ky = physical_a ? hd * By : hd;
    const T kz = physical_a ? hd * Bz : hd;
    const T Ax = (a * Ex + b * (cx + kx)) / static_cast<T>(2.998e8);
    const T Ay = (a * Ey + b * (cy + ky)) / static_cast<T>(2.998e8);
    const T Az = (a * Ez + b * (cz + kz)) / static_cast<T>(2.998e8);
    o[3] = Ax * fr;
    o[7] = Ay * fr;
    o[11] = Az * fz;
}

// ------------------------------------------------------------------ static field painters (K10-K12)

// programCurrentLoopShape (empic.js:295-345): 1000-segment Biot-Savart sum.
template <typename T>
__global__ __launch_bounds__(256) void loop_shape_kernel(T u_R, int nr, int nz, T* __restrict__ out)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nr) * nz) return;
    const int i = static_cast<int>(c % nr), j = static_cast<int>(c / nr);
    const T pi = static_cast<T>(3.14159265359);
    const T constant = u_R * static_cast<T>(0.001) * static_cast<T>(1.25663706e-6) / (static_cast<T>(4.0) * pi);
    const T tx = (static_cast<T>(i) + static_cast<T>(0.5)) / static_cast<T>(nr);
    const T ty = (static_cast<T>(j) + static_cast<T>(0.5)) / static_cast<T>(nz);
    T Bx = 0, Bz = 0;
    for (int k = 0; k < 1000; ++k) {
        const T cosine = cos_(pi * (static_cast<T>(k) + static_cast<T>(0.5)) / static_cast<T>(1000.0));
        const T r = sqrt_(u_R * u_R + tx * tx + ty * ty - static_cast<T>(2.0) * tx * u_R * cosine);
        const T factor = (r > static_cast<T>(0)) ? constant * static_cast<T>(1.0) / (r * r * r) : static_cast<T>(0);
        Bx += ty * factor * cosine;
        Bz += factor * (u_R - tx * cosine);
    }
    out[4 * c] = Bx; out[4 * c + 1] = static_cast<T>(0); out[4 * c + 2] = Bz; out[4 * c + 3] = static_cast<T>(1);
}

// programCurrentLoop (empic.js:349-389) blended ONE,ONE into B (empic.js:1352-1363).
template <typename T>
__global__ __launch_bounds__(256) void current_loop_kernel(T* __restrict__ Bf, const T* __restrict__ half,
                                                           const T* __restrict__ tenth, int nr, int nz, T u_R, T u_Z,
                                                           T u_I)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nr) * nz) return;
    const int i = static_cast<int>(c % nr), j = static_cast<int>(c / nr);
    const T tx = (static_cast<T>(i) + static_cast<T>(0.5)) / static_cast<T>(nr);
    const T ty = (static_cast<T>(j) + static_cast<T>(0.5)) / static_cast<T>(nz);
    const T a = tx / u_R;
    const T b = (ty - u_Z) / u_R;
    const T sgn = (b > static_cast<T>(0)) ? static_cast<T>(1) : ((b < static_cast<T>(0)) ? static_cast<T>(-1) : static_cast<T>(0));
    const T ab = (b < static_cast<T>(0)) ? -b : b;
    const T* t;
    if (a > static_cast<T>(2.0) || b > static_cast<T>(2.0)) // quirk Q6: no abs on b
        t = tenth + 4 * (static_cast<size_t>(ngp(a / static_cast<T>(10.0), nr)) + static_cast<size_t>(nr) * ngp(ab / static_cast<T>(10.0), nz));
    else
        t = half + 4 * (static_cast<size_t>(ngp(a / static_cast<T>(2.0), nr)) + static_cast<size_t>(nr) * ngp(ab / static_cast<T>(2.0), nz));
    T* o = Bf + 4 * c;
    o[0] += (u_I * sgn) * t[0];
    o[1] += (u_I * static_cast<T>(1)) * t[1];
    o[2] += (u_I * static_cast<T>(1)) * t[2];
    o[3] += (u_I * static_cast<T>(1)) * t[3];
}

// programCurrentZ / programBZ / programBTheta (empic.js:392-464, :1380-1411);
// kind 0: line current on the axis, 1: uniform Bz, 2: uniform Btheta (quirk Q7:
// the undefined initial gl_FragColor is taken as 0).
template <typename T>
__global__ __launch_bounds__(256) void add_uniform_kernel(T* __restrict__ Bf, int nr, int nz, int kind, T value)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nr) * nz) return;
    T* o = Bf + 4 * c;
    if (kind == 0) {
        const T tx = (static_cast<T>(c % nr) + static_cast<T>(0.5)) / static_cast<T>(nr);
        o[1] += value * static_cast<T>(1.25663706e-6) / (static_cast<T>(2.0) * static_cast<T>(3.14159265359) * tx);
    } else if (kind == 1) {
        o[2] += value;
    } else {
        o[1] += value;
    }
    o[3] += static_cast<T>(1);
}

// ------------------------------------------------------------------ uploads, read-back

template <typename T>
__global__ __launch_bounds__(256) void init_particles_kernel(ParticleArrays<T> p, size_t n_padded)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_padded) return;
    p.x[i] = p.y[i] = p.z[i] = p.vx[i] = p.vy[i] = p.vz[i] = static_cast<T>(0);
    p.u1[i] = p.u2[i] = p.c1[i] = p.c2[i] = static_cast<T>(0);
    p.alive[i] = 0; // null-data float textures start at 0 (utilities.js:533-539)
    p.id[i] = static_cast<uint32_t>(i);
}

// out.set({position|velocity}) (empic.js:1199-1244): value*factor in double, one
// rounding into T.  The caller's particle i lives in the slot s with id[s] == i.
template <typename T, typename In>
__global__ __launch_bounds__(256) void set_vec3_kernel(const In* __restrict__ aos, size_t chunk_begin, size_t chunk_n,
                                                       double fxy, double fz, T* a, T* b, T* c, uint8_t* alive,
                                                       const uint32_t* __restrict__ id, size_t n, size_t slot0 = 0)
{
    // (id == nullptr: slot s holds the caller's particle s — nothing has been binned yet — and only the slots
    // [slot0, n) of the chunk are visited instead of all of them)
    const size_t s = slot0 + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = id ? id[s] : s;
    if (i < chunk_begin || i >= chunk_begin + chunk_n) return;
    const In* v = aos + 3 * (i - chunk_begin);
    a[s] = static_cast<T>(static_cast<double>(v[0]) * fxy);
    b[s] = static_cast<T>(static_cast<double>(v[1]) * fxy);
    c[s] = static_cast<T>(static_cast<double>(v[2]) * fz);
    if (alive) alive[s] = 1; // position.w = 1 (empic.js:1205)
}

template <typename T>
__global__ __launch_bounds__(256) void set_rand_kernel(const float* __restrict__ aos4, size_t chunk_begin, size_t chunk_n,
                                                       ParticleArrays<T> p, size_t n)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = p.id[s];
    if (i < chunk_begin || i >= chunk_begin + chunk_n) return;
    const float4 v = *reinterpret_cast<const float4*>(aos4 + 4 * (i - chunk_begin));
    p.u1[s] = static_cast<T>(v.x); p.u2[s] = static_cast<T>(v.y);
    p.c1[s] = static_cast<T>(v.z); p.c2[s] = static_cast<T>(v.w);
}

template <typename T, typename Out>
__global__ __launch_bounds__(256) void get_vec3_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       const T* __restrict__ c, const uint32_t* __restrict__ id,
                                                       size_t n, size_t chunk_begin, size_t chunk_n, Out* __restrict__ aos,
                                                       size_t first = 0, size_t stride = 1)
{
    // slot k of the output = the caller's particle first + k * stride (a ranged or sampled read-back); this launch
    // fills slots [chunk_begin, chunk_begin + chunk_n)
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    size_t i = id[s];
    if (i < first || (i - first) % stride) return;
    i = (i - first) / stride;
    if (i < chunk_begin || i >= chunk_begin + chunk_n) return;
    Out* o = aos + 3 * (i - chunk_begin);
    o[0] = static_cast<Out>(a[s]); o[1] = static_cast<Out>(b[s]); o[2] = static_cast<Out>(c[s]);
}

template <typename T>
__global__ __launch_bounds__(256) void get_rand_kernel(ParticleArrays<T> p, size_t n, size_t chunk_begin, size_t chunk_n,
                                                       float* __restrict__ aos4, uint8_t* __restrict__ alive_out,
                                                       int32_t* __restrict__ cells, int nr, int nz)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = p.id[s];
    if (i < chunk_begin || i >= chunk_begin + chunk_n) return;
    const size_t o = i - chunk_begin;
    if (aos4) {
        aos4[4 * o] = static_cast<float>(p.u1[s]); aos4[4 * o + 1] = static_cast<float>(p.u2[s]);
        aos4[4 * o + 2] = static_cast<float>(p.c1[s]); aos4[4 * o + 3] = static_cast<float>(p.c2[s]);
    }
    if (alive_out) alive_out[o] = p.alive[s];
    if (cells) {
        const T r = sqrt_(p.x[s] * p.x[s] + p.y[s] * p.y[s]);
        cells[o] = ngp(r, nr) + nr * ngp(p.z[s], nz);
    }
}

// out.set({E|B|sink_mask}) packing (empic.js:1159-1197, :1246-1260): value[i][j][k]
// -> RGBA texel i + j*nr; ncomp 3 writes xyz and w = 1, ncomp 1 writes red only.
template <typename T, typename In>
__global__ __launch_bounds__(256) void pack_grid_kernel(const In* __restrict__ in, int nr, int nz, int ncomp,
                                                        T* __restrict__ rgba, uint8_t* __restrict__ mask)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= static_cast<size_t>(nr) * nz) return;
    const int i = static_cast<int>(c % nr), j = static_cast<int>(c / nr);
    const In* v = in + static_cast<size_t>(ncomp) * (static_cast<size_t>(i) * nz + j);
    if (ncomp == 3) {
        rgba[4 * c] = static_cast<T>(v[0]); rgba[4 * c + 1] = static_cast<T>(v[1]);
        rgba[4 * c + 2] = static_cast<T>(v[2]); rgba[4 * c + 3] = static_cast<T>(1);
    } else {
        const T red = static_cast<T>(v[0]);
        rgba[4 * c] = red;
        if (mask) mask[c] = (red > static_cast<T>(0.5)) ? 1 : 0; // the test of empic.js:719
    }
}

template <typename T, typename In>
__global__ __launch_bounds__(256) void convert_kernel(const In* __restrict__ in, T* __restrict__ out, size_t n)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = static_cast<T>(in[i]);
}

// R1/R2/R3/A textures as the reference holds them (RGBA, w = 1), rebuilt from coef.
template <typename T, typename Out>
__global__ __launch_bounds__(256) void unpack_coef_kernel(const T* __restrict__ coef, size_t ncell, int row,
                                                          Out* __restrict__ rgba)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const T* o = coef + 12 * c;
    if (row < 3) {
        rgba[4 * c] = static_cast<Out>(o[4 * row]); rgba[4 * c + 1] = static_cast<Out>(o[4 * row + 1]);
        rgba[4 * c + 2] = static_cast<Out>(o[4 * row + 2]);
    } else {
        rgba[4 * c] = static_cast<Out>(o[3]); rgba[4 * c + 1] = static_cast<Out>(o[7]);
        rgba[4 * c + 2] = static_cast<Out>(o[11]);
    }
    rgba[4 * c + 3] = static_cast<Out>(1);
}

// 512x512 (x,y) pairs -> RGBA with zeros in z,w (what inv_cdf_arr holds, empic.js:234).
template <typename T, typename Out>
__global__ __launch_bounds__(256) void unpack_xy_kernel(const T* __restrict__ xy, size_t n, Out* __restrict__ rgba)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n) return;
    rgba[4 * c] = static_cast<Out>(xy[2 * c]); rgba[4 * c + 1] = static_cast<Out>(xy[2 * c + 1]);
    rgba[4 * c + 2] = static_cast<Out>(0); rgba[4 * c + 3] = static_cast<Out>(0);
}

// ------------------------------------------------------------------ binning by cell tile

template <typename T>
__device__ __forceinline__ uint32_t tile_key(T x, T y, T z, int nr, int nz, int ntx, uint32_t garbage)
{
    T r;
    int ic, jc;
    if (!deposit_cell(x, y, z, nr, nz, r, ic, jc)) return garbage;
    return static_cast<uint32_t>(ic / kTileSide) + static_cast<uint32_t>(ntx) * static_cast<uint32_t>(jc / kTileSide);
}

constexpr int kBinPer = 8; // particles per lane in the binning passes

template <typename T>
__global__ __launch_bounds__(256) void bin_count_kernel(ParticleArrays<T> p, size_t n, int nr, int nz, int ntx,
                                                        uint32_t ntiles, uint32_t* __restrict__ tile_count)
{
    extern __shared__ uint32_t hist[];
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) hist[t] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * (256 * kBinPer);
    for (int k = 0; k < kBinPer; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[tile_key(p.x[i], p.y[i], p.z[i], nr, nz, ntx, ntiles - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256)
        if (hist[t]) atomicAdd(&tile_count[t], hist[t]);
}

// Exclusive scan of the tile counts, reset of the cursors, and the scatter's work
// list: each bin is cut into chunks of kDepositChunk particles.  One workgroup.
static __global__ __launch_bounds__(1024) void bin_scan_kernel(const uint32_t* __restrict__ tile_count, uint32_t ntiles,
                                                        uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                                        BlockWork* __restrict__ work, uint32_t* __restrict__ nwork,
                                                        uint32_t chunk = kDepositChunk)
{
    __shared__ uint32_t part_p[1024], part_b[1024];
    const uint32_t per = (ntiles + 1023) / 1024;
    const uint32_t t0 = threadIdx.x * per;
    uint32_t sp = 0, sb = 0;
    for (uint32_t t = t0; t < t0 + per && t < ntiles; ++t) {
        const uint32_t c = tile_count[t];
        sp += c;
        sb += (c + chunk - 1) / chunk;
    }
    part_p[threadIdx.x] = sp;
    part_b[threadIdx.x] = sb;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t ap = 0, ab = 0;
        if (static_cast<int>(threadIdx.x) >= off) { ap = part_p[threadIdx.x - off]; ab = part_b[threadIdx.x - off]; }
        __syncthreads();
        part_p[threadIdx.x] += ap;
        part_b[threadIdx.x] += ab;
        __syncthreads();
    }
    uint32_t run_p = part_p[threadIdx.x] - sp, run_b = part_b[threadIdx.x] - sb;
    for (uint32_t t = t0; t < t0 + per && t < ntiles; ++t) {
        const uint32_t c = tile_count[t];
        tile_start[t] = run_p;
        tile_cursor[t] = 0;
        // The last bin holds the particles that were clipped when binned.  They are
        // scanned too (re-injection can bring them back before the next binning); their
        // LDS window is tile 0's, whatever lands elsewhere takes the global path.
        for (uint32_t b = 0; b * chunk < c; ++b) {
            BlockWork w;
            w.tile = (t + 1 < ntiles) ? t : 0;
            w.begin = run_p + b * chunk;
            w.end = run_p + ((b + 1) * chunk < c ? (b + 1) * chunk : c);
            w.pad = 0;
            work[run_b++] = w;
        }
        run_p += c;
    }
    if (threadIdx.x == 1023) {
        tile_start[ntiles] = part_p[1023];
        *nwork = part_b[1023];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bin_scatter_kernel(ParticleArrays<T> src, ParticleArrays<T> dst, size_t n, int nr,
                                                          int nz, int ntx, uint32_t ntiles,
                                                          const uint32_t* __restrict__ tile_start,
                                                          uint32_t* __restrict__ tile_cursor)
{
    extern __shared__ uint32_t hist[];
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) hist[t] = 0;
    __syncthreads();
    const size_t base = static_cast<size_t>(blockIdx.x) * (256 * kBinPer);
    uint32_t key[kBinPer], rank[kBinPer];
#pragma unroll
    for (int k = 0; k < kBinPer; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        key[k] = 0; rank[k] = 0;
        if (i < n) {
            key[k] = tile_key(src.x[i], src.y[i], src.z[i], nr, nz, ntx, ntiles - 1);
            rank[k] = atomicAdd(&hist[key[k]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256) {
        const uint32_t c = hist[t];
        if (c) hist[t] = tile_start[t] + atomicAdd(&tile_cursor[t], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBinPer; ++k) {
        const size_t i = base + static_cast<size_t>(k) * 256 + threadIdx.x;
        if (i >= n) continue;
        const size_t d = static_cast<size_t>(hist[key[k]]) + rank[k];
        dst.x[d] = src.x[i]; dst.y[d] = src.y[i]; dst.z[d] = src.z[i];
        dst.vx[d] = src.vx[i]; dst.vy[d] = src.vy[i]; dst.vz[d] = src.vz[i];
        dst.u1[d] = src.u1[i]; dst.u2[d] = src.u2[i]; dst.c1[d] = src.c1[i]; dst.c2[d] = src.c2[i];
        dst.alive[d] = src.alive[i];
        dst.id[d] = src.id[i];
    }
}

// First binning of a randomly ordered upload in two levels, each level a scatter staged through LDS.
//
// A single scatter by tile writes every particle to an unrelated place: from a 2048-particle chunk into 1000+ tiles the
// runs are 2 elements long and every wave store touches 64 lines.  Here the population is scattered first by a COARSE
// key (div = ceil(sqrt(ntiles)) consecutive tiles; coarse_start[c] = tile_start[c * div]) and then by tile inside each
// coarse bin, and in both passes a workgroup sorts its 4096-particle chunk in LDS before it writes: every column goes
// through a stage so that the 64 lanes of a wave store 64 consecutive elements of the locally sorted order (one run
// of ~128 elements per bin and chunk).  The data moves twice, but in full lines.
//
// Columns: NT arrays of T (the first three are the position the key is taken from), then an optional byte column and
// the 32-bit id column.  Key(x, y, z) returns the tile, or ~0u for a slot that is not to be copied (a dead slot of the
// decomposed box).  Coarse pass: chunk_first == nullptr, chunks tile [0, n), bin = key / d.  Fine pass: chunk b
// belongs to coarse bin c with chunk_first[c] <= b < chunk_first[c + 1] and covers
// [tile_start[c * div] + m * chunk, tile_start[(c + 1) * div]), bin = key - c * div, d = 1.  Either way the bin's
// destination is tile_start[key_lo + bin * d] + a workgroup-level reservation on cursor[key_lo + bin * d].
constexpr int kSortThreads = 1024, kSortPer = 4, kSortChunk = kSortThreads * kSortPer, kSortMaxBins = 1024;

template <typename T, int NT, bool HAS_BYTE>
struct SortColumns {
    const T* src[NT];
    T* dst[NT];
    const uint8_t* src_byte;
    uint8_t* dst_byte;
    const uint32_t* src_id;
    uint32_t* dst_id;
};

inline size_t sort_scatter_lds(size_t elem) { return sizeof(uint32_t) * (kSortMaxBins + 32 + kSortChunk) + 3 * kSortChunk * elem; }

// chunk_first[c] for the fine pass: exclusive prefix of ceil(count_c / chunk) over the coarse bins (one workgroup)
static __global__ __launch_bounds__(1024) void sort_chunks_kernel(const uint32_t* __restrict__ tile_start, uint32_t ntiles, uint32_t div,
                                                                  uint32_t ncoarse, uint32_t* __restrict__ chunk_first)
{
    __shared__ uint32_t part[1024];
    const uint32_t c = threadIdx.x;
    uint32_t v = 0;
    if (c < ncoarse) {
        const uint32_t lo = tile_start[c * div], hi = tile_start[min((c + 1) * div, ntiles)];
        v = (hi - lo + kSortChunk - 1) / kSortChunk;
    }
    part[c] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t t = c >= static_cast<uint32_t>(o) ? part[c - o] : 0;
        __syncthreads();
        part[c] += t;
        __syncthreads();
    }
    if (c < ncoarse) chunk_first[c] = part[c] - v;
    if (c == ncoarse) chunk_first[c] = part[1023];
}

template <typename T, int NT, bool HAS_BYTE, typename Key>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(SortColumns<T, NT, HAS_BYTE> col, size_t n, Key key_of_pos, uint32_t ntiles, uint32_t div,
                                                                     uint32_t ncoarse, const uint32_t* __restrict__ tile_start,
                                                                     uint32_t* __restrict__ cursor, const uint32_t* __restrict__ chunk_first)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(sort_lds);  // [kSortMaxBins]: counts, then local firsts, then global bases
    uint32_t* wsum = hist + kSortMaxBins;                    // [32]
    uint32_t* dst_of = wsum + 32;                            // [chunk]: global destination of locally sorted position j
    T* stage = reinterpret_cast<T*>(dst_of + kSortChunk);    // [3][chunk]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // which part of the array, which bins
    size_t begin, end;
    uint32_t key_lo, d, nbins;
    if (chunk_first == nullptr) {
        begin = static_cast<size_t>(blockIdx.x) * kSortChunk;
        end = min(n, begin + kSortChunk);
        key_lo = 0; d = div; nbins = ncoarse;
    } else {
        if (blockIdx.x >= chunk_first[ncoarse]) return;
        uint32_t lo = 0, hi = ncoarse; // largest c with chunk_first[c] <= blockIdx.x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (chunk_first[mid] <= blockIdx.x) lo = mid; else hi = mid;
        }
        key_lo = lo * div; d = 1; nbins = min(div, ntiles - key_lo);
        begin = static_cast<size_t>(tile_start[key_lo]) + static_cast<size_t>(blockIdx.x - chunk_first[lo]) * kSortChunk;
        end = min(static_cast<size_t>(tile_start[min(key_lo + div, ntiles)]), begin + kSortChunk);
    }
    hist[tid] = 0;
    __syncthreads();

    T px[kSortPer], py[kSortPer], pz[kSortPer];
    uint32_t bin[kSortPer], rank[kSortPer], pos[kSortPer];
#pragma unroll
    for (int k = 0; k < kSortPer; ++k) {
        const size_t i = begin + static_cast<size_t>(k) * kSortThreads + tid;
        bin[k] = ~0u; rank[k] = 0; px[k] = py[k] = pz[k] = 0;
        if (i < end) {
            px[k] = col.src[0][i]; py[k] = col.src[1][i]; pz[k] = col.src[2][i];
            const uint32_t key = key_of_pos(px[k], py[k], pz[k]);
            if (key != ~0u) {
                bin[k] = (key - key_lo) / d;
                rank[k] = atomicAdd(&hist[bin[k]], 1u);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the (<= 1024) bin counts: local first position of each bin
    const uint32_t cnt = hist[tid];
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const uint32_t w = lane < kSortThreads / 64 ? wsum[lane] : 0;
        uint32_t wi = w;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(wi, o);
            if (lane >= static_cast<uint32_t>(o)) wi += t;
        }
        if (lane < kSortThreads / 64) wsum[lane] = wi - w;
        if (lane == kSortThreads / 64 - 1) wsum[16] = wi;
    }
    __syncthreads();
    hist[tid] = incl - cnt + wsum[wave];
    const uint32_t total = wsum[16];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPer; ++k) pos[k] = bin[k] != ~0u ? hist[bin[k]] + rank[k] : ~0u;
    __syncthreads();
    if (tid < nbins) {
        const uint32_t idx = key_lo + tid * d;
        hist[tid] = cnt ? tile_start[idx] + atomicAdd(&cursor[idx], cnt) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPer; ++k)
        if (pos[k] != ~0u) dst_of[pos[k]] = hist[bin[k]] + rank[k];

    // the columns, three at a time: into the stage at the sorted position, out of it in sorted order
    constexpr int NCOL = NT + (HAS_BYTE ? 1 : 0) + 1;
    uint32_t* stage32 = reinterpret_cast<uint32_t*>(stage);
#pragma unroll
    for (int c0 = 0; c0 < NCOL; c0 += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int c = c0 + s;
            if (c >= NCOL) break;
#pragma unroll
            for (int k = 0; k < kSortPer; ++k) {
                if (pos[k] == ~0u) continue;
                const size_t i = begin + static_cast<size_t>(k) * kSortThreads + tid;
                if (c < NT) {
                    T v;
                    if (c == 0) v = px[k]; else if (c == 1) v = py[k]; else if (c == 2) v = pz[k]; else v = col.src[c < NT ? c : 0][i];
                    stage[s * kSortChunk + pos[k]] = v;
                } else if (HAS_BYTE && c == NT) {
                    stage32[s * kSortChunk * (sizeof(T) / 4) + pos[k]] = col.src_byte[i];
                } else {
                    stage32[s * kSortChunk * (sizeof(T) / 4) + pos[k]] = col.src_id[i];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int c = c0 + s;
            if (c >= NCOL) break;
#pragma unroll
            for (int k = 0; k < kSortPer; ++k) {
                const uint32_t j = static_cast<uint32_t>(k) * kSortThreads + tid;
                if (j >= total) continue;
                const size_t to = dst_of[j];
                if (c < NT) col.dst[c < NT ? c : 0][to] = stage[s * kSortChunk + j];
                else if (HAS_BYTE && c == NT) col.dst_byte[to] = static_cast<uint8_t>(stage32[s * kSortChunk * (sizeof(T) / 4) + j]);
                else col.dst_id[to] = stage32[s * kSortChunk * (sizeof(T) / 4) + j];
            }
        }
        __syncthreads();
    }
}

template <typename T>
struct RzTileKey {
    int nr, nz, ntx;
    uint32_t last;
    __device__ __forceinline__ uint32_t operator()(T x, T y, T z) const { return tile_key(x, y, z, nr, nz, ntx, last); }
};

} // namespace fpic
