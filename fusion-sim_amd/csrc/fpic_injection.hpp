// fpic_injection.hpp — the injection table of out.set({source_pdf}) built on the device
// (SURVEY.md 8(f) next-3).
//
// The reference builds it on the host in JavaScript (empic.js:1263-1339): row-conditional
// CDFs cdf_y[i][j], the marginal cdf_x[i], and their inverses sampled on a 512x512 lattice
// f1 = i/511, f2 = j/511 with linear interpolation inside a bin.  All arithmetic is IEEE
// double in the reference's order (sequential running sums; no FMA contraction in this
// build), the stores into its Float32Array are one rounding to float: the table is
// bit-identical to the reference's (tests/golden/inv_cdf_*.f32.gz).  Quirks kept: an empty
// row gives NaN entries (Q3); a NaN x makes the reference index cdf_y[NaN] and throw (Q12),
// reported through *throws.
#pragma once

#include <hip/hip_runtime.h>

#include "fpic_internal.hpp"

namespace fpic {

// cdf_y: one lane per grid row, running sum over j in the reference's order (empic.js:1272-1283)
template <typename In>
__global__ __launch_bounds__(64) void cdf_rows_kernel(const In* __restrict__ pdf, int nr, int nz, double* __restrict__ row_cdf,
                                                      double* __restrict__ row_sum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr) return;
    const In* p = pdf + static_cast<size_t>(i) * nz;
    double* c = row_cdf + static_cast<size_t>(i) * nz;
    double acc = 0.0;
    for (int j = 0; j < nz; ++j) { acc += static_cast<double>(p[j]); c[j] = acc; }
    for (int j = 0; j < nz; ++j) c[j] /= acc; // 0/0 -> NaN for an empty row (quirk Q3)
    row_sum[i] = acc;
}

// cdf_x: a single lane, running sum over i (empic.js:1285-1291)
__global__ void cdf_cols_kernel(const double* __restrict__ row_sum, int nr, double* __restrict__ col_cdf)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double total = 0.0;
    for (int i = 0; i < nr; ++i) { total += row_sum[i]; col_cdf[i] = total; }
    for (int i = 0; i < nr; ++i) col_cdf[i] /= total;
}

// `while (cdf[a] < f) a++` of inverse_cdf_x / inverse_cdf_y: a comparison with NaN is false
// and stops the scan; reading past the end yields undefined, whose comparison is false too
__device__ __forceinline__ int first_not_below(const double* cdf, int len, double f)
{
    int a = 0;
    while (a < len && cdf[a] < f) ++a;
    return a;
}

// (a + (f - cdf[a-1])/(cdf[a] - cdf[a-1]))/len with the a == 0 and past-the-end cases
// (empic.js:1304-1308, :1321-1325)
__device__ __forceinline__ double interpolate_bin(const double* cdf, int len, int a, double f)
{
    if (a == 0) return (f / cdf[0]) / len;
    if (a >= len) return __builtin_nan("");
    return (a + (f - cdf[a - 1]) / (cdf[a] - cdf[a - 1])) / len;
}

// one lane per lattice point (i, j) (empic.js:1328-1339); table texel (i,j) at i + 512*j
template <typename T>
__global__ __launch_bounds__(256) void inverse_cdf_kernel(const double* __restrict__ col_cdf, const double* __restrict__ row_cdf,
                                                          int nr, int nz, T* __restrict__ table_xy, int* __restrict__ throws)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kCdfSide * kCdfSide) return;
    const int i = t % kCdfSide, j = t / kCdfSide;
    const double f1 = i / static_cast<double>(kCdfSide - 1);
    const double x = interpolate_bin(col_cdf, nr, first_not_below(col_cdf, nr, f1), f1);
    if (x != x) { *throws = 1; return; } // cdf_y[NaN][j]: TypeError in the reference
    const double fl = floor(x * nr);
    const int row = (fl < nr - 1) ? static_cast<int>(fl) : nr - 1; // Math.min(len-1, floor(x*len))
    if (row < 0) { *throws = 1; return; }
    const double* rc = row_cdf + static_cast<size_t>(row) * nz;
    const double f2 = j / static_cast<double>(kCdfSide - 1);
    const double y = interpolate_bin(rc, nz, first_not_below(rc, nz, f2), f2);
    table_xy[2 * static_cast<size_t>(t)] = static_cast<T>(static_cast<float>(x));      // Float32Array store
    table_xy[2 * static_cast<size_t>(t) + 1] = static_cast<T>(static_cast<float>(y));
}

} // namespace fpic
