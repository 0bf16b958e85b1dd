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

#ifdef _OPENMP
#include <omp.h>
void es3d_set_threads(int n) { omp_set_num_threads(n); }
#endif

#define REAL float
#define SUF _f32
#define FN_FLOOR floorf
#include "es3d_oracle_impl.h"
#undef REAL
#undef SUF
#undef FN_FLOOR

#define REAL double
#define SUF _f64
#define FN_FLOOR floor
#include "es3d_oracle_impl.h"
#undef REAL
#undef SUF
#undef FN_FLOOR
