// host_tables_test.cpp — csrc/fpic_host.cpp (the reference's host-side tables: derived constants, toFixed(20) shader literals,
// the 11x11 stamp) built for the HOST and checked against the numbers SURVEY.md 8(c) records from the reference's own
// JavaScript under Node (stamp centre 0.042796533554792404, (4,5) 0.03870982676744461, corners 0, sum 0.99999999 in fp32).
// Run under AddressSanitizer + UBSan by `make -C fusion-sim_amd sanitize`.
#include "../../fusion-sim_amd/csrc/fpic_host.cpp"

#include <cmath>
#include <cstdio>

int main()
{
    using namespace fpic;
    float w[kStampCells];
    build_stamp(w);
    double sum = 0;
    for (float v : w) sum += v;
    int bad = 0;
    auto near = [&](double got, double want, double tol, const char* what) {
        if (!(std::fabs(got - want) <= tol)) { std::printf("%s: %.17g, expected %.17g\n", what, got, want); ++bad; }
    };
    near(w[5 + 11 * 5], 0.042796533554792404, 1e-9, "stamp centre");
    near(w[4 + 11 * 5], 0.03870982676744461, 1e-9, "stamp (4,5)");
    near(w[0], 0.0, 0.0, "stamp corner");
    near(w[10 + 11 * 10], 0.0, 0.0, "stamp corner");
    near(sum, 1.0, 1e-6, "stamp sum");
    for (int j = 0; j < 11; ++j)
        for (int i = 0; i < 11; ++i)
            if (w[i + 11 * j] != w[(10 - i) + 11 * j] || w[i + 11 * j] != w[j + 11 * i]) { std::printf("stamp asymmetric at %d %d\n", i, j); ++bad; }
    // toFixed(20): twenty decimals are re-read; values below 5e-21 read back as zero, large ones exactly
    near(shader_literal(0.1), 0.1, 1e-20, "literal 0.1");
    near(shader_literal(1e-25), 0.0, 0.0, "literal 1e-25");
    near(shader_literal(-2.5e-3), -2.5e-3, 1e-20, "literal -2.5e-3");
    if (shader_literal(1.5e300) != 1.5e300 || shader_literal(-1.7e308) != -1.7e308) { std::printf("large literal\n"); ++bad; }
    fpic_spec s{};
    s.radius = 0.5; s.height = 2.0; s.dt = 2e-9; s.particle_mass = 1.67e-27; s.particle_charge = 1.602e-19;
    const Constants k = derive_constants(s);
    near(k.h, 1.602e-19 * 2e-9 / (2 * 1.67e-27), 1e-12, "h");
    near(k.step_factor, 2e-9 * 2.998e8, 0.0, "dt c");
    near(k.f_rz, 4.0, 0.0, "factor_r / factor_z");
    near(k.f_zr, 0.25, 0.0, "factor_z / factor_r");
    std::printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
