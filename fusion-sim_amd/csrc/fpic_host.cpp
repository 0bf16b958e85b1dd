// fpic_host.cpp — host-side tables of the particle pusher factory and of set().
//
// These are the parts of the reference that run in JavaScript on the host, not in
// shaders: the derived constants (empic.js:44-46, :852), the shader literals
// printed with toFixed(20) (empic.js:23-25), the 11x11 stamp (empic.js:949-971)
// and the inverse-CDF injection table (empic.js:1263-1339).  All arithmetic is
// IEEE double, as in JavaScript; stores into the reference's Float32Arrays are a
// single rounding to float.
#include "fpic_internal.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

namespace fpic {

Constants derive_constants(const fpic_spec& s)
{
    const double speed_of_light = 2.998e8; // empic.js:27
    Constants k;
    k.h = s.particle_charge * s.dt / (2 * s.particle_mass); // empic.js:44
    k.factor_r = 1 / s.radius;                               // empic.js:45
    k.factor_z = 1 / s.height;                               // empic.js:46
    k.step_factor = s.dt * speed_of_light;                   // empic.js:852, :927
    k.f_rz = k.factor_r / k.factor_z;                        // empic.js:527, :566
    k.f_zr = k.factor_z / k.factor_r;                        // empic.js:606
    return k;
}

// N(x) = x.toFixed(20): the decimal text a GLSL compiler re-reads (empic.js:23-25).
double shader_literal(double x)
{
    char buf[512];
    std::snprintf(buf, sizeof buf, "%.20f", x);
    return std::strtod(buf, nullptr);
}

// empic.js:949-971.  Weights are cos^2(pi*d/10) inside radius 5, normalised by the
// sum of the float-rounded values; index i + 11*j, red channel only.
void build_stamp(float w[kStampCells])
{
    const int ns = kStampSide;
    const double mid = (ns - 1) / 2.0;
    double sum = 0.0;
    for (int j = 0; j < ns; ++j) {
        for (int i = 0; i < ns; ++i) {
            const double d = std::sqrt(std::pow(i - mid, 2) + std::pow(j - mid, 2));
            const double c = std::cos(0.5 * 3.141592653589793 * d / mid);
            const float v = static_cast<float>(std::pow(c > 0.0 ? c : 0.0, 2));
            w[i + ns * j] = v;
            sum += v;
        }
    }
    for (int k = 0; k < kStampCells; ++k) w[k] = static_cast<float>(w[k] / sum);
}

namespace {

// First index a in [from, len) with cdf[a] >= f, scanning upwards; a comparison with
// NaN is false and stops the scan, exactly like the reference's while loop.
inline int first_not_below(const double* cdf, int len, int from, double f)
{
    int a = from;
    while (a < len && cdf[a] < f) ++a;
    return a;
}

// (a + (f - cdf[a-1])/(cdf[a] - cdf[a-1]))/len with the a == 0 and past-the-end
// cases of inverse_cdf_x / inverse_cdf_y (empic.js:1304-1308, :1321-1325).
inline double interpolate_bin(const double* cdf, int len, int a, double f)
{
    if (a == 0) return (f / cdf[0]) / len;
    if (a >= len) return std::numeric_limits<double>::quiet_NaN();
    return (a + (f - cdf[a - 1]) / (cdf[a] - cdf[a - 1])) / len;
}

} // namespace

// out.set({source_pdf}) (empic.js:1263-1339).  pdf is value[i][j] flattened; table
// receives kCdfSide*kCdfSide (x, y) pairs, texel (i,j) at i + kCdfSide*j.  Returns
// false where the reference throws (x = NaN makes it index cdf_y[NaN]).
bool build_inverse_cdf(const double* pdf, int nr, int nz, std::vector<float>& table_xy)
{
    std::vector<double> row_cdf(static_cast<size_t>(nr) * nz);
    std::vector<double> col_cdf(nr);
    double total = 0.0;
    for (int i = 0; i < nr; ++i) {
        double* rc = &row_cdf[static_cast<size_t>(i) * nz];
        double acc = 0.0;
        for (int j = 0; j < nz; ++j) { acc += pdf[static_cast<size_t>(i) * nz + j]; rc[j] = acc; }
        for (int j = 0; j < nz; ++j) rc[j] /= acc; // 0/0 -> NaN for an empty row (quirk Q3)
        total += acc;
        col_cdf[i] = total;
    }
    for (int i = 0; i < nr; ++i) col_cdf[i] /= total;

    table_xy.assign(static_cast<size_t>(2) * kCdfSide * kCdfSide, 0.0f);
    int a = 0;
    for (int i = 0; i < kCdfSide; ++i) {
        const double f1 = i / static_cast<double>(kCdfSide - 1);
        // col_cdf is non-decreasing when finite, so the scan may resume where it stopped
        a = (col_cdf[0] == col_cdf[0]) ? first_not_below(col_cdf.data(), nr, a, f1) : 0;
        const double x = interpolate_bin(col_cdf.data(), nr, a, f1);
        if (x != x) return false;
        const double fl = std::floor(x * nr);
        const int row = (fl < nr - 1) ? static_cast<int>(fl) : nr - 1; // Math.min(len-1, floor(x*len))
        if (row < 0) return false;
        const double* rc = &row_cdf[static_cast<size_t>(row) * nz];
        int b = 0;
        for (int j = 0; j < kCdfSide; ++j) {
            const double f2 = j / static_cast<double>(kCdfSide - 1);
            b = (rc[0] == rc[0]) ? first_not_below(rc, nz, b, f2) : 0;
            const double y = interpolate_bin(rc, nz, b, f2);
            const size_t t = static_cast<size_t>(i) + static_cast<size_t>(kCdfSide) * j;
            table_xy[2 * t] = static_cast<float>(x);
            table_xy[2 * t + 1] = static_cast<float>(y);
        }
    }
    return true;
}

} // namespace fpic
