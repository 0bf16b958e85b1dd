// ablate_wave_private.hip — development probe (round 3, VERDICT r02 "missing 6"): north_star names "per-wavefront
// privatised accumulators ... wavefront shuffle reductions" for the scatter; what is built is ONE accumulator window per
// workgroup with LDS atomics (ds_add_f64).  This measures the alternatives on the same synthetic scatter as
// scripts/ablate_lds_atomics.hip (1e8 elements, 4 channels per element, random cell inside the window, 256 threads):
//   shared      one window per workgroup, ds_add_f64                                  (the product's form)
//   private     one window per WAVEFRONT (4 copies, each a quarter of the cells so that the LDS footprint is the same),
//               ds_add_f64 inside the copy, copies summed at the end
//   shuffle     one window per wavefront and NO atomics: lanes that hit the same cell are found with ballots, their values
//               are summed across the wave with shuffles, the first of them does a plain read-modify-write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ idx, size_t n, double* out, int cells)
{
    extern __shared__ unsigned char raw[];
    double* tile = reinterpret_cast<double*>(raw);
    const int copies = MODE == 0 ? 1 : 4, per_copy = cells / copies;
    for (int i = threadIdx.x; i < cells * 4; i += 256) tile[i] = 0.0;
    __syncthreads();
    double* mine = tile + (MODE == 0 ? 0 : (threadIdx.x >> 6) * per_copy * 4);
    const size_t per = 32768, b = blockIdx.x * per;
    for (size_t i = b + threadIdx.x; i < b + per && i < n; i += 256) {
        const unsigned c = idx[i] % per_copy;
        double* t = mine + 4 * c;
        if (MODE < 2) {
            atomicAdd(t, 1.0); atomicAdd(t + 1, 2.0); atomicAdd(t + 2, 3.0); atomicAdd(t + 3, 4.0);
        } else {
            // in-wave combination: groups of lanes with the same cell, one group per round
            double v0 = 1.0, v1 = 2.0, v2 = 3.0, v3 = 4.0;
            unsigned long long todo = __ballot(1);
            const int lane = threadIdx.x & 63;
            while (todo) {
                const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                const unsigned c0 = __shfl(c, leader);
                const unsigned long long same = __ballot(c == c0) & todo;
                if (c == c0) {
                    // sum over the lanes of `same` (usually one lane: cells are random inside the window)
                    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                    for (unsigned long long m = same; m; m &= m - 1) {
                        const int l = __ffsll(static_cast<long long>(m)) - 1;
                        s0 += __shfl(v0, l); s1 += __shfl(v1, l); s2 += __shfl(v2, l); s3 += __shfl(v3, l);
                    }
                    if (lane == leader) { t[0] += s0; t[1] += s1; t[2] += s2; t[3] += s3; }
                }
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = threadIdx.x; i < per_copy * 4; i += 256) {
        double s = 0.0;
        for (int w = 0; w < copies; ++w) s += tile[w * per_copy * 4 + i];
        acc += s;
    }
    if (acc == 123457.0) out[0] = acc;
}
__global__ void fill(unsigned* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    unsigned long long z = (i + 77) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
    p[i] = (unsigned)(z >> 20);
}
template <int MODE> float run(const unsigned* idx, size_t n, double* out, int cells) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((n + 32767) / 32768);
    const size_t sh = (size_t)cells * 4 * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    k<MODE><<<nb, 256, sh>>>(idx, n, out, cells); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<MODE><<<nb, 256, sh>>>(idx, n, out, cells);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
int main() {
    const size_t n = 100000000ull; unsigned* idx; double* out;
    CK(hipMalloc((void**)&idx, n * 4)); CK(hipMalloc((void**)&out, 64));
    fill<<<(n + 255) / 256, 256>>>(idx, n); CK(hipDeviceSynchronize());
    const int cells = 48 * 48;
    printf("# scripts/ablate_wave_private.hip on 1x MI355X: 4 double adds per element, 1e8 elements, %d accumulator cells of LDS per\n"
           "# workgroup (48x48 window, 73.7 KB), 256-thread workgroups, 32768 elements each; 0.4 GB index stream.\n", cells);
    printf("shared window, ds_add_f64 (the product's form)                       %.3f ms\n", run<0>(idx, n, out, cells));
    printf("one window per wavefront (quarter of the cells each), ds_add_f64     %.3f ms\n", run<1>(idx, n, out, cells));
    printf("one window per wavefront, ballot + shuffle combination, no atomics   %.3f ms\n", run<2>(idx, n, out, cells));
    return 0;
}
