/*
 * pic_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see pic_oracle.h for what is
 * and is not pinned by the reference).  Build: oracle/Makefile
 *     gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC pic_oracle.c -lm
 */
#include "pic_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
/* libpic_oracle_omp.so only: thread count of the all-cores timing variant */
void orc_set_threads(int n) { omp_set_num_threads(n); }
#endif

/* ---- host-side routines of the factory and of out.set() ---- */

/* empic.js:27 (speed_of_light), :44 (h), :45-46 (factor_r, factor_z), :852 (u_step_factor). */
void orc_constants(double radius, double height, double dt, double mass, double charge, double out[6])
{
    const double speed_of_light = 2.998e8;
    out[0] = charge * dt / (2 * mass);
    out[1] = 1 / radius;
    out[2] = 1 / height;
    out[3] = dt * speed_of_light;
    out[4] = out[1] / out[2];
    out[5] = out[2] / out[1];
}

/* empic.js:23-25: shader literals are printed with toFixed(20) and re-read by the
 * GLSL compiler. */
double orc_tofixed20(double x)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%.20f", x);
    return strtod(buf, NULL);
}

/* empic.js:949-971.  shape_arr is a Float32Array: each store rounds to float, the
 * running sum reads the rounded value back as a double. */
void orc_stamp(float out[121])
{
    const int nshape = ORC_NSHAPE;
    const double mid = (nshape - 1) / 2.0;
    const double pi = 3.141592653589793;
    double sum = 0;
    for (int j = 0; j < nshape; ++j)
        for (int i = 0; i < nshape; ++i) {
            double d = sqrt(pow(i - mid, 2) + pow(j - mid, 2));
            double c = cos(0.5 * pi * d / mid);
            out[i + nshape * j] = (float)pow(c > 0.0 ? c : 0.0, 2);
            sum += out[i + nshape * j];
        }
    for (int k = 0; k < nshape * nshape; ++k) out[k] = (float)(out[k] / sum);
}

/* empic.js:1263-1339, all arithmetic in double as in JavaScript, stores into a
 * Float32Array.  Reading past the end of a JS array yields undefined, whose
 * comparison is false and whose arithmetic is NaN; indexing cdf_y[NaN] throws. */
int orc_inv_cdf(const double* pdf, int nr, int nz, float* out)
{
    double* cdf_y = (double*)malloc(sizeof(double) * (size_t)nr * nz);
    double* cdf_x = (double*)malloc(sizeof(double) * (size_t)nr);
    if (!cdf_y || !cdf_x) { free(cdf_y); free(cdf_x); return -3; }
    double sum_x = 0;
    for (int i = 0; i < nr; ++i) {
        double sum_y = 0;
        for (int j = 0; j < nz; ++j) { sum_y += pdf[(size_t)i * nz + j]; cdf_y[(size_t)i * nz + j] = sum_y; }
        for (int j = 0; j < nz; ++j) cdf_y[(size_t)i * nz + j] /= sum_y;
        sum_x += sum_y;
        cdf_x[i] = sum_x;
    }
    for (int i = 0; i < nr; ++i) cdf_x[i] /= sum_x;

    for (size_t k = 0; k < (size_t)4 * ORC_N_CDF * ORC_N_CDF; ++k) out[k] = 0.0f;
    int rc = 0;
    for (int i = 0; i < ORC_N_CDF && rc == 0; ++i) {
        double f1 = i / 511.0;
        /* inverse_cdf_x (empic.js:1293-1309) */
        double x;
        {
            int a = 0;
            while (a < nr && cdf_x[a] < f1) a++;
            if (a == 0) x = (f1 / cdf_x[0]) / nr;
            else if (a == nr) x = NAN;
            else x = (a + (f1 - cdf_x[a - 1]) / (cdf_x[a] - cdf_x[a - 1])) / nr;
        }
        if (x != x) { rc = -1; break; } /* cdf_y[NaN][j] -> TypeError in the reference */
        /* row used by inverse_cdf_y (empic.js:1312) */
        double fl = floor(x * nr);
        int row = (fl < nr - 1) ? (int)fl : nr - 1;
        if (row < 0) { rc = -1; break; }
        const double* cy = cdf_y + (size_t)row * nz;
        for (int j = 0; j < ORC_N_CDF; ++j) {
            double f2 = j / 511.0;
            double y;
            int b = 0;
            while (b < nz && cy[b] < f2) b++;
            if (b == 0) y = (f2 / cy[0]) / nz;
            else if (b == nz) y = NAN;
            else y = (b + (f2 - cy[b - 1]) / (cy[b] - cy[b - 1])) / nz;
            out[4 * ((size_t)i + (size_t)j * ORC_N_CDF)] = (float)x;
            out[4 * ((size_t)i + (size_t)j * ORC_N_CDF) + 1] = (float)y;
        }
    }
    free(cdf_y);
    free(cdf_x);
    return rc;
}

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator of the RNG
 * extension mode; constants of the Random123 reference implementation. */
void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ---- per-fragment arithmetic, float and double ---- */

#define REAL float
#define FN(x) orc_f32_##x
#define SQRT sqrtf
#define COS cosf
#define FLOOR floorf
#include "pic_oracle_impl.h"
#undef REAL
#undef FN
#undef SQRT
#undef COS
#undef FLOOR

#define REAL double
#define FN(x) orc_f64_##x
#define SQRT sqrt
#define COS cos
#define FLOOR floor
#include "pic_oracle_impl.h"
#undef REAL
#undef FN
#undef SQRT
#undef COS
#undef FLOOR
