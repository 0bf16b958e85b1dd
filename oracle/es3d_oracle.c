/*
 * es3d_oracle.c — CPU oracle of the CART3D electrostatic mode (BASELINE.json configs[2..4]):
 * CIC deposit, FFT Poisson solve, CIC gather + Boris push on a periodic 3-D box.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
 * PARITY UNPINNED — no reference counterpart exists (see es3d_oracle_impl.h); the
 * mode is anchored by analytic known answers in tests/test_oracle_es3d.py instead
 * (single Fourier mode of the discrete Poisson operator, cold-plasma oscillation at
 * omega_p, exact charge conservation, Boris gyration angle 2 atan(h|B|)).
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ES3D_EPS0 8.8541878128e-12
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- 3-D complex FFT in double: radix-2 when the length is a power of two, a plain DFT otherwise ---- */
static void fft_line(double* re, double* im, int n, int sign, double* wr, double* wi, double* tr, double* ti)
{
    if ((n & (n - 1)) == 0) {
        for (int i = 1, j = 0; i < n; ++i) {
            int bit = n >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
        }
        for (int len = 2; len <= n; len <<= 1) {
            const int step = n / len;
            for (int i = 0; i < n; i += len)
                for (int k = 0; k < len / 2; ++k) {
                    const double cr = wr[k * step], ci = sign * wi[k * step];
                    const int a = i + k, b = i + k + len / 2;
                    const double xr = re[b] * cr - im[b] * ci, xi = re[b] * ci + im[b] * cr;
                    re[b] = re[a] - xr; im[b] = im[a] - xi;
                    re[a] += xr; im[a] += xi;
                }
        }
        return;
    }
    for (int k = 0; k < n; ++k) {
        double sr = 0, si = 0;
        for (int m = 0; m < n; ++m) {
            const int idx = (int)(((long long)k * m) % n);
            const double cr = wr[idx], ci = sign * wi[idx];
            sr += re[m] * cr - im[m] * ci;
            si += re[m] * ci + im[m] * cr;
        }
        tr[k] = sr; ti[k] = si;
    }
    memcpy(re, tr, sizeof(double) * n);
    memcpy(im, ti, sizeof(double) * n);
}

/* in place, unnormalised, sign = -1 forward / +1 inverse; index i + nx*(j + ny*k) */
void es3d_fft3(double* re, double* im, int nx, int ny, int nz, int sign)
{
    const int dims[3] = { nx, ny, nz };
    const size_t stride[3] = { 1, (size_t)nx, (size_t)nx * ny };
    for (int ax = 0; ax < 3; ++ax) {
        const int n = dims[ax];
        double* wr = (double*)malloc(sizeof(double) * n * 6);
        double *wi = wr + n, *lr = wi + n, *li = lr + n, *tr = li + n, *ti = tr + n;
        for (int k = 0; k < n; ++k) { wr[k] = cos(2 * M_PI * k / n); wi[k] = sin(2 * M_PI * k / n); }
        const int o1 = (ax + 1) % 3, o2 = (ax + 2) % 3;
        for (int b = 0; b < dims[o2]; ++b)
            for (int a = 0; a < dims[o1]; ++a) {
                const size_t base = a * stride[o1] + b * stride[o2];
                for (int m = 0; m < n; ++m) { lr[m] = re[base + m * stride[ax]]; li[m] = im[base + m * stride[ax]]; }
                fft_line(lr, li, n, sign, wr, wi, tr, ti);
                for (int m = 0; m < n; ++m) { re[base + m * stride[ax]] = lr[m]; im[base + m * stride[ax]] = li[m]; }
            }
        free(wr);
    }
}

/* eigenvalues of minus the 3-point second difference on n periodic nodes of spacing d */
double* es3d_k2_table(int n, double d)
{
    double* t = (double*)malloc(sizeof(double) * n);
    for (int l = 0; l < n; ++l) {
        const double s = 2.0 / d * sin(M_PI * l / n);
        t[l] = s * s;
    }
    return t;
}

/* ---- charge-conserving current deposit in integers (full EM mode) ----
 *
 * Coordinates are DOUBLED fixed-point lattice coordinates: H = 2 * (cell * 2^14 + w1), S = 2^15 per cell (the same
 * 14-bit quantised positions the charge deposit uses, doubled so that midpoints stay integers).  The move a -> b
 * (nearest periodic image) is cut at a relay point (zigzag, Umeda et al. 2003) into two segments that each stay
 * inside one cell; per segment and cell the Villasenor-Buneman fluxes through the four dual faces of each direction,
 *     Jx(j+b, k+c) += dX * [ 3 * Ay_b * Az_c  +-  dY dZ ],   Ay_0 = 2S - Y1 - Y2,  Ay_1 = Y1 + Y2   (+ for b == c),
 * in units where one particle crossing a whole face carries 12 * S^3 * ... = 96 * 2^42: with the charge grid of
 * es3d_deposit (2^42 per particle) the lattice continuity equation holds EXACTLY in integers,
 *     96 * (rho_fixed^(n+1) - rho_fixed^n)[node] + (div Jfix)[node] = 0.
 * Jfix holds 3 int64 per node: the x-edge (i+1/2,j,k), the y-edge, the z-edge. */
static inline int64_t floor_div(int64_t a, int64_t s) { return a >= 0 ? a / s : -((-a + s - 1) / s); }

static void current_segment(const int64_t p1[3], const int64_t p2[3], const int64_t cell[3], int nx, int ny, int nz, int Z, int64_t* Jfix)
{
    const int64_t S = 32768;
    int64_t l1[3], l2[3], d[3], A0[3], A1[3];
    for (int m = 0; m < 3; ++m) {
        l1[m] = p1[m] - cell[m] * S; l2[m] = p2[m] - cell[m] * S;
        d[m] = l2[m] - l1[m];
        A1[m] = l1[m] + l2[m]; A0[m] = 2 * S - A1[m];
    }
    const int n[3] = { nx, ny, nz };
    int64_t c0[3], c1[3];
    for (int m = 0; m < 3; ++m) {
        c0[m] = ((cell[m] % n[m]) + n[m]) % n[m];
        c1[m] = (c0[m] + 1 == n[m]) ? 0 : c0[m] + 1;
    }
    for (int m = 0; m < 3; ++m) {                      /* direction of the current */
        if (d[m] == 0) continue;
        const int u = (m + 1) % 3, v = (m + 2) % 3;  /* the two transverse axes */
        const int64_t cross = d[u] * d[v];
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) {
                const int64_t flux = d[m] * (3 * (b ? A1[u] : A0[u]) * (c ? A1[v] : A0[v]) + (b == c ? cross : -cross)) * Z;
                int64_t idx[3];
                idx[m] = c0[m]; idx[u] = b ? c1[u] : c0[u]; idx[v] = c ? c1[v] : c0[v];
                Jfix[3 * ((size_t)idx[0] + (size_t)nx * ((size_t)idx[1] + (size_t)ny * idx[2])) + m] += flux;
            }
    }
}

void es3d_current(const int64_t a[3], const int64_t b_in[3], int nx, int ny, int nz, int Z, int64_t* Jfix)
{
    const int64_t S = 32768;
    const int n[3] = { nx, ny, nz };
    int64_t b[3], ca[3], cb[3], r[3];
    for (int m = 0; m < 3; ++m) {
        const int64_t box = (int64_t)n[m] * S;
        int64_t dd = b_in[m] - a[m];
        if (2 * dd > box) dd -= box;                /* nearest periodic image of the end point */
        else if (2 * dd < -box) dd += box;
        b[m] = a[m] + dd;
        ca[m] = floor_div(a[m], S);
        cb[m] = floor_div(b[m], S);
        /* relay point: the midpoint inside one cell, the face between two (both coordinates are even: the midpoint is an integer) */
        r[m] = (ca[m] == cb[m]) ? (a[m] + b[m]) / 2 : (ca[m] > cb[m] ? ca[m] : cb[m]) * S;
    }
    current_segment(a, r, ca, nx, ny, nz, Z, Jfix);
    current_segment(r, b, cb, nx, ny, nz, Z, Jfix);
}

#ifdef _OPENMP
#include <omp.h>
void es3d_set_threads(int n) { omp_set_num_threads(n); }
#endif

#define REAL float
#define SUF _f32
#define FN_FLOOR floorf
#define FN_FMA fmaf
#include "es3d_oracle_impl.h"
#undef REAL
#undef SUF
#undef FN_FLOOR
#undef FN_FMA

#define REAL double
#define SUF _f64
#define FN_FLOOR floor
#define FN_FMA fma
#include "es3d_oracle_impl.h"
#undef REAL
#undef SUF
#undef FN_FLOOR
#undef FN_FMA
