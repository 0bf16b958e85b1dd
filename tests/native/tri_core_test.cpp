// Host build of the arithmetic core of fusion-sim_amd/csrc/fes_tri.hpp (the interface solve of the slab-decomposed Poisson
// solve): P ranks of m planes each run down_sweep, exchange (y_1, y_m), solve the interface system and run up_sweep; the
// result is written for tests/test_tri_core.py, which compares it with numpy's FFT solve of the same periodic system.
//   tri_core_test <P> <m> <nmodes> <float|double> <in.bin> <out.bin>
// in.bin: lam[nmodes] (double) then f[P*m][nmodes] complex double; out.bin: phi[P*m][nmodes] complex double
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../fusion-sim_amd/csrc/fes_tri.hpp"

template <typename T>
int run(int P, int m, int nm, const std::vector<double>& lam, const std::vector<double>& f, std::vector<double>& out)
{
    const int nz = P * m;
    const double pi = 3.14159265358979323846;
    double tw[festri::kMaxRanks][2] = {};
    for (int k = 0; k < P; ++k) { tw[k][0] = std::cos(2 * pi * k / P); tw[k][1] = std::sin(2 * pi * k / P); }
    for (int q = 0; q < nm; ++q) {
        const festri::Mode md = festri::make_mode(lam[q], m);
        std::vector<T> col(2 * static_cast<size_t>(nz));
        for (int z = 0; z < nz; ++z) { col[2 * z] = static_cast<T>(f[2 * (static_cast<size_t>(z) * nm + q)]); col[2 * z + 1] = static_cast<T>(f[2 * (static_cast<size_t>(z) * nm + q) + 1]); }
        double y1[festri::kMaxRanks][2] = {}, ym[festri::kMaxRanks][2] = {};
        for (int r = 0; r < P; ++r) {
            festri::down_sweep<T>(md, m, col.data() + 2 * static_cast<size_t>(r) * m, 1, 1.0, y1[r][0], y1[r][1], ym[r][0], ym[r][1]);
            // the exchanged planes are stored in T
            for (int c = 0; c < 2; ++c) { y1[r][c] = static_cast<double>(static_cast<T>(y1[r][c])); ym[r][c] = static_cast<double>(static_cast<T>(ym[r][c])); }
        }
        for (int r = 0; r < P; ++r) {
            double ar, ai, br, bi;
            festri::interface_values(md, P, r, y1, ym, tw, ar, ai, br, bi);
            festri::up_sweep<T>(md, m, col.data() + 2 * static_cast<size_t>(r) * m, 1, ar, ai, br, bi);
        }
        for (int z = 0; z < nz; ++z) { out[2 * (static_cast<size_t>(z) * nm + q)] = col[2 * z]; out[2 * (static_cast<size_t>(z) * nm + q) + 1] = col[2 * z + 1]; }
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 5 && !std::strcmp(argv[1], "zero")) { // zero <n> <in.bin> <out.bin>: the singular line
        const int n = std::atoi(argv[2]);
        std::vector<double> f(n), phi(n);
        FILE* fi = std::fopen(argv[3], "rb");
        if (!fi || std::fread(f.data(), 8, n, fi) != static_cast<size_t>(n)) return 2;
        std::fclose(fi);
        festri::zero_line_serial(f.data(), n, phi.data());
        FILE* fo = std::fopen(argv[4], "wb");
        std::fwrite(phi.data(), 8, n, fo);
        std::fclose(fo);
        return 0;
    }
    if (argc != 7) return 1;
    const int P = std::atoi(argv[1]), m = std::atoi(argv[2]), nm = std::atoi(argv[3]);
    if (P < 1 || P > festri::kMaxRanks) return 1;
    std::vector<double> lam(nm), f(2 * static_cast<size_t>(P) * m * nm), out(f.size());
    FILE* fi = std::fopen(argv[5], "rb");
    if (!fi || std::fread(lam.data(), 8, nm, fi) != static_cast<size_t>(nm) || std::fread(f.data(), 8, f.size(), fi) != f.size()) return 2;
    std::fclose(fi);
    const int rc = !std::strcmp(argv[4], "float") ? run<float>(P, m, nm, lam, f, out) : run<double>(P, m, nm, lam, f, out);
    FILE* fo = std::fopen(argv[6], "wb");
    std::fwrite(out.data(), 8, out.size(), fo);
    std::fclose(fo);
    return rc;
}
