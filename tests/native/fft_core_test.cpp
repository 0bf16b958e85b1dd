// Host build of the arithmetic core of csrc/fes_fft.hpp (radix-2/4/8 butterflies, Stockham index maps): every supported
// length, both directions, both precisions against a direct O(N^2) transform in long double.  The device kernels wrap
// exactly these functions; here the workgroup is emulated by "all butterflies of a pass load, then all store".
#include "../../fusion-sim_amd/csrc/fes_fft.hpp"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

using fesfft::C2;

template <typename T, int R, bool INV>
void run_pass(std::vector<C2<T>>& col, const std::vector<C2<T>>& tw, int N, int Ns)
{
    const int per = N / R;
    std::vector<C2<T>> regs(static_cast<size_t>(per) * R);
    for (int j = 0; j < per; ++j) {
        C2<T> v[R];
        fesfft::pass_load<T, R, INV>(col.data(), tw.data(), N, Ns, j, v);
        for (int r = 0; r < R; ++r) regs[static_cast<size_t>(j) * R + r] = v[r];
    }
    for (int j = 0; j < per; ++j) {
        C2<T> v[R];
        for (int r = 0; r < R; ++r) v[r] = regs[static_cast<size_t>(j) * R + r];
        fesfft::pass_store<T, R>(col.data(), Ns, j, v);
    }
}

template <typename T, bool INV>
double check(int logn, unsigned seed)
{
    const int N = 1 << logn;
    std::mt19937 gen(seed);
    std::uniform_real_distribution<double> u(-1, 1);
    std::vector<C2<T>> col(fesfft::swz_len(N) + 1), tw(N);
    std::vector<long double> xr(N), xi(N);
    for (int n = 0; n < N; ++n) { const C2<T> v{ static_cast<T>(u(gen)), static_cast<T>(u(gen)) }; col[fesfft::swz(n)] = v; xr[n] = v.x; xi[n] = v.y; }
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int t = 0; t < N; ++t) tw[t] = { static_cast<T>(std::cos(-2 * pi * t / N)), static_cast<T>(std::sin(-2 * pi * t / N)) };
    int Ns = 1;
    for (int rem = logn; rem > 0;) {
        const int rl = fesfft::next_radix_log(rem);
        if (rl == 3) run_pass<T, 8, INV>(col, tw, N, Ns);
        else if (rl == 2) run_pass<T, 4, INV>(col, tw, N, Ns);
        else run_pass<T, 2, INV>(col, tw, N, Ns);
        Ns <<= rl;
        rem -= rl;
    }
    double worst = 0, scale = 0;
    for (int k = 0; k < N; ++k) {
        long double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            const long double a = (INV ? 2 : -2) * pi * (static_cast<long double>((static_cast<long long>(n) * k) % N)) / N;
            const long double c = std::cos(a), s = std::sin(a);
            sr += xr[n] * c - xi[n] * s;
            si += xr[n] * s + xi[n] * c;
        }
        worst = std::fmax(worst, std::fmax(std::fabs(static_cast<double>(sr - col[fesfft::swz(k)].x)), std::fabs(static_cast<double>(si - col[fesfft::swz(k)].y))));
        scale = std::fmax(scale, std::fmax(std::fabs(static_cast<double>(sr)), std::fabs(static_cast<double>(si))));
    }
    return worst / scale;
}

int main()
{
    int bad = 0;
    for (int logn = 1; logn <= 10; ++logn) {
        const double f0 = check<float, false>(logn, 1 + logn), f1 = check<float, true>(logn, 11 + logn);
        const double d0 = check<double, false>(logn, 21 + logn), d1 = check<double, true>(logn, 31 + logn);
        std::printf("N=%4d  float fwd %.2e inv %.2e   double fwd %.2e inv %.2e\n", 1 << logn, f0, f1, d0, d1);
        if (f0 > 2e-6 || f1 > 2e-6 || d0 > 4e-15 || d1 > 4e-15) ++bad;
    }
    std::printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
