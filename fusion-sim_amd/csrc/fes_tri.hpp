// fes_tri.hpp — the decomposed direction of the slab-decomposed Poisson solve WITHOUT transposing the spectrum.
//
// No reference counterpart (the reference has no field solve in its step loop, empic.js:1436-1505):
// PARITY UNPINNED; the definition is oracle/es3d_oracle_impl.h's es3d_poisson (phi_hat = rho_hat / (eps0 K^2), K^2 the
// eigenvalues of the three-point Laplacian, mean mode 0), met within the solve's tolerance like the transform path.
//
// K^2 = k2x + k2y + k2z with k2z the eigenvalues of the three-point second difference along z.  After the x and y
// transforms of a rank's own planes, every mode (kx, ky) therefore obeys a periodic TRIDIAGONAL system along z,
//
//     -phi[z-1] + (2 + lam) phi[z] - phi[z+1] = f[z],   lam = (k2x + k2y) dz^2,  f = rho_hat dz^2 / (eps0 nx ny),
//
// which the ranks solve by substructuring instead of transposing 2 x nz/P planes' worth of spectrum per rank through an
// all-to-all (250 MB per rank and sub-step at 512^3 on 8 ranks: what the links of a strong-scaling run would carry):
//
//   1. down sweep (tri_down_kernel, one thread per mode, z sequential, planes coalesced across modes): forward
//      elimination of the rank's own m = nz/P equations with homogeneous ends, g_j = (f_j + g_{j-1}) c_j, stored in place;
//      its last value is y_m = (T^-1 f)_m, and y_1 = (T^-1 f)_1 = sum_j v_j f_j is accumulated on the way.
//      T = tridiag(-1, 2 + lam, -1) of size m; everything about it is known in closed form from r = exp(-theta),
//      2 cosh(theta) = 2 + lam:  c_j = r (1 - r^2j) / (1 - r^(2j+2)),  v_j = (T^-1)_{j1} = r^j (1 - r^(2(m+1-j))) / D,
//      w_j = (T^-1)_{jm} = v_{m+1-j},  D = 1 - r^(2(m+1)).
//   2. all-gather of (y_1, y_m): TWO planes of the half spectrum per rank instead of m (1/32 of the transposes' bytes at
//      m = 64), plus the m values of the (0, 0) mode's line.
//   3. up sweep (tri_up_kernel): every rank solves the 2P-unknown interface system of its mode redundantly — it is
//      block-circulant over the ranks, so a P-point DFT over the ranks turns it into P 2 x 2 systems —
//          x_r - beta x_{r+1} - alpha z_{r-1} = y_1^r,   z_r - alpha x_{r+1} - beta z_{r-1} = y_m^r
//      (x_r, z_r = phi on the rank's first / last plane; alpha = v_1 = w_m, beta = v_m = w_1), takes a = z_{r-1} and
//      b = x_{r+1}, and substitutes back: phi_j = y_j + a v_j + b w_j with y_j = g_j + c_j y_{j+1}, in place.
//   The mode (0, 0) is singular (lam = 0; the mean mode is dropped, as in the transform path): its whole line is gathered
//   and every rank solves it with two prefix sums (tri_zero_line).
//
// Arithmetic is double for both precisions of the state: the local problems of the long waves are ill-conditioned
// (cond ~ (2 (m+1) / pi)^2) and a float recurrence misses the float transform path's 2e-5 by a factor of three, while
// double recurrences on float storage stay at 1e-7.  The exchanged planes and the stored g are T.  Every difference
// 1 - r^(2k) and the interface system's determinant at rank-frequency 0 — which cancel like theta for the long waves —
// are formed without cancellation (one_minus, det0): against numpy's transform solve the core is within 2e-14 for lam from 1e-9
// to 8 on generic right-hand sides, and within 100 eps (1 + 1 / lam) always — the system's condition number is 4 / lam and a
// zero-mean right-hand side of a nearly singular mode picks up rounding along the constant vector, which the transform (it
// divides that component by lam exactly) does not: 2e-13 for every mode of a cubic-cell 512-grid (tests/test_tri_core.py).
//
// The arithmetic core below is plain C++ that also compiles for the host: tests/test_tri_core.py builds it with g++ and
// checks it against numpy's FFT solve.
#pragma once

#include <cmath>
#include <cstddef>

#if defined(__HIPCC__)
#define FESTRI_HD __host__ __device__ __forceinline__
#else
#define FESTRI_HD inline
#endif

namespace festri {

constexpr int kMaxRanks = 8;        // the interface system is solved in registers: 2 x kMaxRanks complex doubles per mode
constexpr double kTiny = 1e-280;    // below this a power of r is as good as zero (and must not be divided by)

struct Mode {
    double r, r2, rinv, log2r, lnr;  // exp(-theta), its square, 1 / r, log2(r), ln(r)
    double D;                        // 1 - r^(2(m+1))
    double alpha, beta;              // (T^-1)_11 = (T^-1)_mm, (T^-1)_1m = (T^-1)_m1
    double det0;                     // (1 - beta)^2 - alpha^2, the interface system's determinant at rank-frequency 0
};

// 1 - r^(2k) without cancellation: p = r^(2k) as the caller has it (accurate RELATIVE to itself); where it is not small
// the difference is taken from the exponent instead (long waves: 2 k theta < ln 2)
FESTRI_HD double one_minus(const Mode& q, double p, int k) { return p < 0.5 ? 1.0 - p : -expm1(2.0 * k * q.lnr); }

FESTRI_HD Mode make_mode(double lam, int m)
{
    Mode q;
    const double d = 2.0 + lam, s = sqrt(lam * (lam + 4.0));    // sqrt(d^2 - 4) without cancellation
    q.r = 2.0 / (d + s);                            // the root of r^2 - d r + 1 inside the unit circle
    const double one_r = (lam + s) / (d + s);       // 1 - r
    q.r2 = q.r * q.r;
    q.rinv = 1.0 / q.r;
    q.lnr = -log1p(one_r * q.rinv);
    q.log2r = q.lnr * 1.4426950408889634074;
    const double rm = exp2(m * q.log2r);            // r^m (0 when it underflows: then it IS nothing)
    const double em = one_minus(q, rm * rm, m), em1 = one_minus(q, rm * rm * q.r2, m + 1);
    q.D = em1;
    q.alpha = q.r * em / em1;
    q.beta = rm * one_r * (1.0 + q.r) / em1;
    // 1 - alpha - beta = (1 - r)(1 - r^m)(1 - r^(m+1)) / D: three factors that each cancel like theta, taken one by one
    const double f_m = rm < 0.5 ? 1.0 - rm : -expm1(m * q.lnr), f_m1 = rm * q.r < 0.5 ? 1.0 - rm * q.r : -expm1((m + 1) * q.lnr);
    q.det0 = one_r * f_m * f_m1 / em1 * (1.0 - q.beta + q.alpha);
    return q;
}

// c_j = r (1 - r^(2j)) / (1 - r^(2j+2)) of the forward elimination for a given p = r^(2j)
FESTRI_HD double elim_c(const Mode& q, double p, int j) { return q.r * one_minus(q, p, j) / one_minus(q, p * q.r2, j + 1); }

// r^j for the sweep that walks j downwards: one multiplication by 1 / r, or the exponential when the previous value had
// underflowed (a power that has become 0 would stay 0 while the true one grows back into range)
FESTRI_HD double power_down(const Mode& q, double prev, int j) { return prev > kTiny ? prev * q.rinv : exp2(j * q.log2r); }

// 2 x 2 system of rank-frequency k of the interface system:
//   (1 - beta w) X - alpha conj(w) Z = Y1,   -alpha w X + (1 - beta conj(w)) Z = Ym,   w = exp(2 pi i k / P)
FESTRI_HD void solve_pair(const Mode& q, double wr, double wi, double y1r, double y1i, double ymr, double ymi, double& xr, double& xi, double& zr, double& zi)
{
    const double a11r = 1.0 - q.beta * wr, a11i = -q.beta * wi;
    const double a12r = -q.alpha * wr, a12i = q.alpha * wi;       // -alpha conj(w)
    const double a21r = -q.alpha * wr, a21i = -q.alpha * wi;      // -alpha w
    const double a22r = 1.0 - q.beta * wr, a22i = q.beta * wi;
    // a11 a22 - a12 a21 = |1 - beta w|^2 - alpha^2 = det0 + 2 beta (1 - cos): real, a sum of non-negative terms
    const double det = q.det0 + 2.0 * q.beta * (1.0 - wr);
    const double inv = 1.0 / det;
    // X = (Y1 a22 - a12 Ym) / det, Z = (a11 Ym - a21 Y1) / det
    xr = ((y1r * a22r - y1i * a22i) - (a12r * ymr - a12i * ymi)) * inv;
    xi = ((y1r * a22i + y1i * a22r) - (a12r * ymi + a12i * ymr)) * inv;
    zr = ((a11r * ymr - a11i * ymi) - (a21r * y1r - a21i * y1i)) * inv;
    zi = ((a11r * ymi + a11i * ymr) - (a21r * y1i + a21i * y1r)) * inv;
}

// The values a = phi on the last plane of the slab below and b = phi on the first plane of the slab above, from every
// rank's (y_1, y_m): y1[r], ym[r] as (re, im) pairs, tw[k] = (cos, sin)(2 pi k / P).
FESTRI_HD void interface_values(const Mode& q, int P, int rank, const double (&y1)[kMaxRanks][2], const double (&ym)[kMaxRanks][2], const double (&tw)[kMaxRanks][2],
                                double& ar, double& ai, double& br, double& bi)
{
    ar = ai = br = bi = 0.0;
    const int below = (rank + P - 1) % P, above = (rank + 1) % P;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < kMaxRanks; ++k) {
        if (k >= P) continue;
        double s1r = 0, s1i = 0, smr = 0, smi = 0;           // Y1_k = sum_r y1[r] conj(w)^(k r), likewise Ym_k
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int r = 0; r < kMaxRanks; ++r) {
            if (r >= P) continue;
            const int t = (k * r) % P;
            const double c = tw[t][0], s = -tw[t][1];
            s1r += y1[r][0] * c - y1[r][1] * s; s1i += y1[r][0] * s + y1[r][1] * c;
            smr += ym[r][0] * c - ym[r][1] * s; smi += ym[r][0] * s + ym[r][1] * c;
        }
        double xr, xi, zr, zi;
        solve_pair(q, tw[k][0], tw[k][1], s1r, s1i, smr, smi, xr, xi, zr, zi);
        const int ta = (k * below) % P, tb = (k * above) % P;  // z_below = (1/P) sum_k Z_k w^(k below), x_above likewise
        ar += zr * tw[ta][0] - zi * tw[ta][1]; ai += zr * tw[ta][1] + zi * tw[ta][0];
        br += xr * tw[tb][0] - xi * tw[tb][1]; bi += xr * tw[tb][1] + xi * tw[tb][0];
    }
    const double inv = 1.0 / P;
    ar *= inv; ai *= inv; br *= inv; bi *= inv;
}

// One mode's down sweep over the rank's m planes: f (scaled by `scale`) -> g in place; returns y_1 and y_m.
// col[j * stride] is the complex value of plane j (pairs of T).
template <typename T>
FESTRI_HD void down_sweep(const Mode& q, int m, T* col, size_t stride, double scale, double& y1r, double& y1i, double& ymr, double& ymi)
{
    double gr = 0, gi = 0, p = 1.0, rj = 1.0;
    double pk = exp2(2.0 * (m + 1) * q.log2r);               // r^(2k), k = m + 1 - j, walked downwards with j
    double ej = one_minus(q, q.r2, 1);                       // 1 - r^(2j); each step's denominator is the next step's numerator
    const double invD = 1.0 / q.D;
    y1r = y1i = 0;
#if defined(__HIPCC__)
#pragma unroll 8
#endif
    for (int j = 1; j <= m; ++j) {
        T* at = col + 2 * (static_cast<size_t>(j - 1) * stride);
        const double fr = static_cast<double>(at[0]) * scale, fi = static_cast<double>(at[1]) * scale;
        p *= q.r2; rj *= q.r;                                // r^(2j), r^j
        const double ej1 = one_minus(q, p * q.r2, j + 1);
        const double c = q.r * ej / ej1;                     // c_j = r (1 - r^(2j)) / (1 - r^(2j+2))
        ej = ej1;
        gr = (fr + gr) * c; gi = (fi + gi) * c;
        pk = pk > kTiny ? pk * q.rinv * q.rinv : exp2(2.0 * (m + 1 - j) * q.log2r);
        const double v = rj * one_minus(q, pk, m + 1 - j) * invD;     // (T^-1)_{1j} = r^j (1 - r^(2(m+1-j))) / D
        y1r += v * fr; y1i += v * fi;
        at[0] = static_cast<T>(gr); at[1] = static_cast<T>(gi);
    }
    ymr = gr; ymi = gi;
}

// One mode's up sweep: g -> phi_hat in place, given a (the plane below the slab) and b (the plane above it).
template <typename T>
FESTRI_HD void up_sweep(const Mode& q, int m, T* col, size_t stride, double ar, double ai, double br, double bi)
{
    double yr = 0, yi = 0, rk = 1.0;                         // y_{j+1}; r^k with k = m + 1 - j
    double rj = exp2((m + 1) * q.log2r);                     // r^(j+1) to start with
    double ej1 = q.D;                                        // 1 - r^(2(j+1))
    const double invD = 1.0 / q.D;
#if defined(__HIPCC__)
#pragma unroll 8
#endif
    for (int j = m; j >= 1; --j) {
        T* at = col + 2 * (static_cast<size_t>(j - 1) * stride);
        rj = power_down(q, rj, j);
        rk *= q.r;
        const double gr = static_cast<double>(at[0]), gi = static_cast<double>(at[1]);
        const double ej = one_minus(q, rj * rj, j);
        if (j == m) { yr = gr; yi = gi; }
        else { const double c = q.r * ej / ej1; yr = gr + c * yr; yi = gi + c * yi; }
        ej1 = ej;
        const double v = rj * one_minus(q, rk * rk, m + 1 - j) * invD;   // (T^-1)_{j1}
        const double w = rk * ej * invD;                                 // (T^-1)_{jm}
        at[0] = static_cast<T>(yr + ar * v + br * w);
        at[1] = static_cast<T>(yi + ai * v + bi * w);
    }
}

// The singular line of the mode (0, 0): -phi[z-1] + 2 phi[z] - phi[z+1] = f[z] - mean(f) on the whole periodic line of
// n values, mean(phi) = 0.  With s[z] = phi[z+1] - phi[z]:  s[z] = s0 - G[z], G the inclusive prefix sum of f - mean,
// s0 = mean(G) (periodicity), phi[z] = sum_{t<z} s[t] - mean.  Serial form (the host test's, and the definition).
inline void zero_line_serial(const double* f, int n, double* phi)
{
    double mean = 0;
    for (int z = 0; z < n; ++z) mean += f[z];
    mean /= n;
    double G = 0, sumG = 0;
    for (int z = 0; z < n; ++z) { G += f[z] - mean; sumG += G; }
    const double s0 = sumG / n;
    double acc = 0, tot = 0;
    G = 0;
    for (int z = 0; z < n; ++z) { phi[z] = acc; tot += acc; G += f[z] - mean; acc += s0 - G; }
    tot /= n;
    for (int z = 0; z < n; ++z) phi[z] -= tot;
}

} // namespace festri

#if defined(__HIPCC__)

namespace festri {

// geometry of a rank's spectrum: [m planes][ny rows][pitch complex], nxh of a row's values are modes
struct Slab {
    int m, ny, nxh, pitch;
};

// 1. down sweep.  mine = [2][ny][pitch] complex T (y_1, y_m); line0 = [m] complex T, the (0, 0) mode's scaled values.
template <typename T>
__global__ __launch_bounds__(256) void tri_down_kernel(T* __restrict__ hat, Slab s, const double* __restrict__ k2x, const double* __restrict__ k2y, double dz2,
                                                       double scale, T* __restrict__ mine, T* __restrict__ line0)
{
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int j = static_cast<int>(t / s.pitch), i = static_cast<int>(t % s.pitch);
    if (j >= s.ny || i >= s.nxh) return;
    const size_t plane = static_cast<size_t>(s.ny) * s.pitch, at = static_cast<size_t>(j) * s.pitch + i;
    if (i == 0 && j == 0) { // the singular mode: hand its line to the gather as it is (scaled)
        for (int z = 0; z < s.m; ++z) {
            line0[2 * z] = static_cast<T>(static_cast<double>(hat[2 * (z * plane)]) * scale);
            line0[2 * z + 1] = static_cast<T>(static_cast<double>(hat[2 * (z * plane) + 1]) * scale);
        }
        mine[0] = mine[1] = mine[2 * plane] = mine[2 * plane + 1] = static_cast<T>(0);
        return;
    }
    const Mode q = make_mode((k2x[i] + k2y[j]) * dz2, s.m);
    double y1r, y1i, ymr, ymi;
    down_sweep<T>(q, s.m, hat + 2 * at, plane, scale, y1r, y1i, ymr, ymi);
    mine[2 * at] = static_cast<T>(y1r); mine[2 * at + 1] = static_cast<T>(y1i);
    mine[2 * (plane + at)] = static_cast<T>(ymr); mine[2 * (plane + at) + 1] = static_cast<T>(ymi);
}

// 3. up sweep.  all = [P][block] with block = 2 planes (y_1, y_m) then the rank's line of the (0, 0) mode.
template <typename T>
__global__ __launch_bounds__(256) void tri_up_kernel(T* __restrict__ hat, Slab s, const double* __restrict__ k2x, const double* __restrict__ k2y, double dz2,
                                                     const T* __restrict__ all, size_t block, int P, int rank)
{
    __shared__ double tw_s[kMaxRanks][2];
    if (threadIdx.x < kMaxRanks) {
        double sn = 0, cs = 1;
        if (static_cast<int>(threadIdx.x) < P) sincospi(2.0 * threadIdx.x / P, &sn, &cs);
        tw_s[threadIdx.x][0] = cs; tw_s[threadIdx.x][1] = sn;
    }
    __syncthreads();
    const size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int j = static_cast<int>(t / s.pitch), i = static_cast<int>(t % s.pitch);
    if (j >= s.ny || i >= s.nxh) return;
    if (i == 0 && j == 0) return;                           // the singular line has its own kernel
    const size_t plane = static_cast<size_t>(s.ny) * s.pitch, at = static_cast<size_t>(j) * s.pitch + i;
    double y1[kMaxRanks][2], ym[kMaxRanks][2], tw[kMaxRanks][2];
#pragma unroll
    for (int r = 0; r < kMaxRanks; ++r) {
        tw[r][0] = tw_s[r][0]; tw[r][1] = tw_s[r][1];
        y1[r][0] = y1[r][1] = ym[r][0] = ym[r][1] = 0;
        if (r < P) {
            const T* b = all + 2 * (static_cast<size_t>(r) * block);
            y1[r][0] = static_cast<double>(b[2 * at]); y1[r][1] = static_cast<double>(b[2 * at + 1]);
            ym[r][0] = static_cast<double>(b[2 * (plane + at)]); ym[r][1] = static_cast<double>(b[2 * (plane + at) + 1]);
        }
    }
    const Mode q = make_mode((k2x[i] + k2y[j]) * dz2, s.m);
    double ar, ai, br, bi;
    interface_values(q, P, rank, y1, ym, tw, ar, ai, br, bi);
    up_sweep<T>(q, s.m, hat + 2 * at, plane, ar, ai, br, bi);
}

// The (0, 0) line: one workgroup, the whole line of n = P m values in LDS, two inclusive scans (Hillis-Steele, n <= 1024).
template <typename T>
__global__ __launch_bounds__(1024) void tri_zero_line_kernel(T* __restrict__ hat, Slab s, const T* __restrict__ all, size_t block, int P, int rank)
{
    __shared__ double a[2][1024];
    const int n = P * s.m, z = threadIdx.x;
    const size_t plane = static_cast<size_t>(s.ny) * s.pitch;
    for (int comp = 0; comp < 2; ++comp) {
        double f = 0;
        if (z < n) f = static_cast<double>(all[2 * (static_cast<size_t>(z / s.m) * block + 2 * plane + (z % s.m)) + comp]);
        // block-wide sum -> mean
        auto total = [&](double v) -> double {
            __syncthreads();
            a[0][z] = z < n ? v : 0.0;
            __syncthreads();
            for (int k = 512; k > 0; k >>= 1) { if (z < k) a[0][z] += a[0][z + k]; __syncthreads(); }
            const double r = a[0][0];
            __syncthreads();
            return r;
        };
        auto scan = [&](double v) -> double { // inclusive prefix sum over z
            int cur = 0;
            a[0][z] = z < n ? v : 0.0;
            __syncthreads();
            for (int k = 1; k < 1024; k <<= 1) {
                a[cur ^ 1][z] = a[cur][z] + (z >= k ? a[cur][z - k] : 0.0);
                cur ^= 1;
                __syncthreads();
            }
            const double r = a[cur][z];
            __syncthreads();
            return r;
        };
        const double mean = total(f) / n;
        const double G = scan(f - mean);
        const double s0 = total(G) / n;
        const double incl = scan(z < n ? s0 - G : 0.0);       // sum_{t<=z} s[t]
        const double phi = incl - (z < n ? s0 - G : 0.0);     // sum_{t<z} s[t]
        const double tot = total(phi) / n;
        if (z >= rank * s.m && z < (rank + 1) * s.m) hat[2 * (static_cast<size_t>(z - rank * s.m) * plane) + comp] = static_cast<T>(phi - tot);
    }
}

} // namespace festri

#endif
