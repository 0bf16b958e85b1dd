// fpic_host.cpp — host-side tables of the particle pusher factory and of set().
//
// These are the parts of the reference that run in JavaScript on the host, not in
// shaders: the derived constants (empic.js:44-46, :852), the shader literals
// printed with toFixed(20) (empic.js:23-25) and the 11x11 stamp (empic.js:949-971); the
// inverse-CDF injection table is built on the device (fpic_injection.hpp).  All arithmetic is
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

} // namespace fpic
