// fsor_kernels.hpp — gfx950 kernels of the dense iterative solver (matrix_webgl.js:35-711).
//
// One matrix row = one vh x vh block of RGBA texels in the reference (T = vh^2 texels, L = 4T
// columns).  The reference reduces a row with n_power passes that each add 2x2 texels per colour
// channel, then adds the four channels: a radix-4 tree over the block's Morton (Z-order) index
// m = interleave(vx, vy), each node summed as ((c3 + c2) + c1) + c0.  Float addition is not
// associative, so the kernel keeps exactly that tree; what it is free to choose is where the
// operands live.
//
// Storage: a row is handled by one 64-lane wavefront.  Lane l owns the Morton-contiguous chunk
// m in [l*S, (l+1)*S), S = T/W texels, W = min(64, T) lanes, and texel (lane l, slot i) of row r
// is stored at float4 index (r*S + i)*W + l.  Every load instruction of a wave therefore reads
// W consecutive float4 = 1 KB, rows stream from HBM exactly once per product, and the vector is
// kept in the same (slot, lane) order so it is read the same way (from L2).  The lane folds its
// chunk in registers, the top three tree levels are wave shuffles.
//
// Roofline: HBM.  Algorithmic bytes per product = 4 L^2 (the iteration matrix once).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace fsor {

constexpr int kWave = 64;
constexpr int kRowsPerBlock = 4; // one wave per row

struct Shape {
    int n_power, vh, T, W, S, levels_local, levels_wave;
    uint32_t L;
};

__host__ __device__ inline Shape make_shape(int n_power)
{
    Shape s;
    s.n_power = n_power;
    s.vh = 1 << n_power;
    s.T = s.vh * s.vh;
    s.W = s.T < kWave ? s.T : kWave;
    s.S = s.T / s.W;
    s.levels_wave = s.T < kWave ? n_power : 3;      // log4(W)
    s.levels_local = n_power - s.levels_wave;       // log4(S)
    s.L = 4u * static_cast<uint32_t>(s.T);
    return s;
}

// even bits of m -> vx, odd bits -> vy (m < 2^16 * 2^16)
__host__ __device__ inline uint32_t compact_even_bits(uint32_t m)
{
    m &= 0x55555555u;
    m = (m | (m >> 1)) & 0x33333333u;
    m = (m | (m >> 2)) & 0x0F0F0F0Fu;
    m = (m | (m >> 4)) & 0x00FF00FFu;
    m = (m | (m >> 8)) & 0x0000FFFFu;
    return m;
}
__host__ __device__ inline uint32_t spread_bits(uint32_t v)
{
    v &= 0x0000FFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// texel index q = vx + vh*vy of Morton index m, and back
__host__ __device__ inline uint32_t texel_of_morton(uint32_t m, int vh) { return compact_even_bits(m) + vh * compact_even_bits(m >> 1); }
__host__ __device__ inline uint32_t morton_of_texel(uint32_t q, int vh) { return spread_bits(q % vh) | (spread_bits(q / vh) << 1); }
// storage slot (float4 index within a row / within the vector) of Morton index m
__host__ __device__ inline uint32_t slot_of_morton(uint32_t m, const Shape& s) { return (m % s.S) * s.W + m / s.S; }

// Which vector element the sum of matrix row `row` lands in.  Reference (matrix_webgl.js:389-424):
// the summed texture holds row bx + 2vh*by at texel (bx,by); result texel (X,Y) takes channels
// from (2X,2Y), (2X+1,2Y), (2X,2Y+1), (2X+1,2Y+1).
__host__ __device__ inline uint32_t element_of_row(uint32_t row, int vh, bool natural)
{
    if (natural) return row;
    const uint32_t bx = row % (2 * vh), by = row / (2 * vh);
    return 4 * ((bx >> 1) + vh * (by >> 1)) + (bx & 1) + 2 * (by & 1);
}

__device__ inline float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ inline float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// programR (matrix_webgl.js:222-262) into the storage order above.  One thread per stored texel.
__global__ void build_iteration_matrix_kernel(const float* __restrict__ A, float4* __restrict__ R, Shape s, float omega, int scale)
{
    const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; // (row*S + i)*W + l
    const size_t per_row = static_cast<size_t>(s.T);
    if (idx >= per_row * s.L) return;
    const uint32_t row = static_cast<uint32_t>(idx / per_row);
    const uint32_t in_row = static_cast<uint32_t>(idx % per_row);
    const uint32_t m = (in_row % s.W) * s.S + in_row / s.W;
    const uint32_t col = 4 * texel_of_morton(m, s.vh);
    const float* a = A + static_cast<size_t>(row) * s.L;
    const float d = a[row];
    const float4 v = *reinterpret_cast<const float4*>(a + col);
    float4 r;
    r.x = (row == col) ? 0.0f : -v.x / d;
    r.y = (row == col + 1) ? 0.0f : -v.y / d;
    r.z = (row == col + 2) ? 0.0f : -v.z / d;
    r.w = (row == col + 3) ? 0.0f : -v.w / d;
    if (scale) r = make_float4(omega * r.x, omega * r.y, omega * r.z, omega * r.w);
    R[idx] = r;
}

// programC (matrix_webgl.js:266-301), natural element order.
__global__ void build_constant_kernel(const float* __restrict__ A, const float* __restrict__ b, float* __restrict__ C, uint32_t L,
                                      float omega, int scale)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L) return;
    const float v = b[e] / A[static_cast<size_t>(e) * L + e];
    C[e] = scale ? omega * v : v;
}

// natural element order -> (slot, lane) order of the vector (float4 = one texel)
__global__ void permute_vector_kernel(const float4* __restrict__ x, float4* __restrict__ xs, Shape s)
{
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= static_cast<uint32_t>(s.T)) return;
    xs[slot_of_morton(morton_of_texel(q, s.vh), s)] = x[q];
}

// One product: x_out = (sum over the row's tree of R*x) + C (+ keep*x_in), programMVproduct +
// the n_power summing passes + programResult (matrix_webgl.js:305-424).  LOCAL = log4(S).
template <int LOCAL>
__global__ void __launch_bounds__(kWave* kRowsPerBlock)
    product_kernel(const float4* __restrict__ R, const float4* __restrict__ xs_in, const float* __restrict__ x_in,
                   const float* __restrict__ C, float* __restrict__ x_out, float* __restrict__ xs_out, Shape s, float keep,
                   int relaxed, int natural)
{
    const uint32_t lane = threadIdx.x % kWave;
    const uint32_t row = blockIdx.x * kRowsPerBlock + threadIdx.x / kWave;
    if (row >= s.L) return; // whole wave
    const bool active = lane < static_cast<uint32_t>(s.W);
    const float4* r = R + static_cast<size_t>(row) * s.T + lane;
    const float4* xv = xs_in + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);

    if (active) {
        if constexpr (LOCAL == 0) {
            v = mul4(r[0], xv[0]);
        } else {
            // radix-4 counter over the lane's S = 4^LOCAL texels, visited in descending order: a
            // node's children arrive as c3, c2, c1, c0 and are added in that order
            float4 acc[LOCAL];
            const int groups = s.S / 4;
            for (int g = groups - 1; g >= 0; --g) {
                const size_t o = static_cast<size_t>(4 * g) * s.W;
                const float4 p3 = mul4(r[o + 3 * static_cast<size_t>(s.W)], xv[o + 3 * static_cast<size_t>(s.W)]);
                const float4 p2 = mul4(r[o + 2 * static_cast<size_t>(s.W)], xv[o + 2 * static_cast<size_t>(s.W)]);
                const float4 p1 = mul4(r[o + static_cast<size_t>(s.W)], xv[o + static_cast<size_t>(s.W)]);
                const float4 p0 = mul4(r[o], xv[o]);
                v = add4(add4(add4(p3, p2), p1), p0);
#pragma unroll
                for (int lev = 1; lev < LOCAL; ++lev) {
                    const int d = (g >> (2 * (lev - 1))) & 3;
                    acc[lev] = (d == 3) ? v : add4(acc[lev], v);
                    if (d != 0) break;
                    v = acc[lev];
                }
            }
        }
    }
    // top levels across lanes: groups of 4 lanes at stride 4^k
    for (int k = 0, stride = 1; k < s.levels_wave; ++k, stride *= 4) {
        const int base = static_cast<int>(lane) - static_cast<int>((lane / stride) % 4) * stride;
        float4 a, t;
#define FSOR_SHFL4(dst, src, from)                                                                  \
    dst.x = __shfl(src.x, from);                                                                    \
    dst.y = __shfl(src.y, from);                                                                    \
    dst.z = __shfl(src.z, from);                                                                    \
    dst.w = __shfl(src.w, from)
        FSOR_SHFL4(a, v, base + 3 * stride);
        FSOR_SHFL4(t, v, base + 2 * stride);
        a = add4(a, t);
        FSOR_SHFL4(t, v, base + stride);
        a = add4(a, t);
        FSOR_SHFL4(t, v, base);
        v = add4(a, t);
#undef FSOR_SHFL4
    }
    if (lane == 0) {
        const float sum = ((v.x * 1.0f + v.y * 1.0f) + v.z * 1.0f) + v.w * 1.0f;
        const uint32_t e = element_of_row(row, s.vh, natural != 0);
        float out = sum + C[e];
        if (relaxed) out = out + keep * x_in[e];
        x_out[e] = out;
        xs_out[4 * slot_of_morton(morton_of_texel(e / 4, s.vh), s) + (e & 3)] = out;
    }
}

// programStats (matrix_webgl.js:428-452), one thread per vector texel.
__global__ void stats_kernel(const float4* __restrict__ x1, const float4* __restrict__ x2, float4* __restrict__ stats, uint32_t texels)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= texels) return;
    const float4 a = x1[t], b = x2[t];
    float4 o;
    o.x = (((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w) * 0.25f;
    o.y = (((a.x * a.x + a.y * a.y) + a.z * a.z) + a.w * a.w) * 0.25f;
    o.z = (((b.x * b.x + b.y * b.y) + b.z * b.z) + b.w * b.w) * 0.25f;
    const float d0 = fabsf(b.x - a.x), d1 = fabsf(b.y - a.y), d2 = fabsf(b.z - a.z), d3 = fabsf(b.w - a.w);
    float m = d0 < d1 ? d1 : d0;
    m = m < d2 ? d2 : m;
    m = m < d3 ? d3 : m;
    o.w = m;
    stats[t] = o;
}

// storage order -> the reference's texture layout of R (block (bx,by) = row bx + 2vh*by), for read-back
__global__ void export_iteration_matrix_kernel(const float4* __restrict__ R, float4* __restrict__ tex, Shape s)
{
    const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; // nx + mh*ny
    const size_t mh = 2 * static_cast<size_t>(s.T);
    if (idx >= mh * mh) return;
    const uint32_t nx = static_cast<uint32_t>(idx % mh), ny = static_cast<uint32_t>(idx / mh);
    const uint32_t row = nx / s.vh + 2 * s.vh * (ny / s.vh);
    const uint32_t q = nx % s.vh + s.vh * (ny % s.vh);
    tex[idx] = R[static_cast<size_t>(row) * s.T + slot_of_morton(morton_of_texel(q, s.vh), s)];
}

} // namespace fsor
