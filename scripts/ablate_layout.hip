// Round 5 probe: does the LAYOUT of the particle arrays bound the tiled pushes?  profiles/r03_c3_ablation.txt: the
// electrostatic push with every LDS access removed streams at 5.0 TB/s where a float4 copy does 6.3.  Its workgroups read and
// write six arrays (structure of arrays, 16 bytes per lane and array): twelve streams per workgroup.  Here the same in-place
// update (read six scalars, a few FMAs, write six) over 5e8 particles in float
//   soa      six arrays of n, workgroup w takes the slots [w * chunk, (w + 1) * chunk) of each (the library's layout)
//   aosoa B  blocks of B particles, a block = x[B] y[B] z[B] vx[B] vy[B] vz[B] contiguous: a wave reads 6 KB in one piece (B = 256),
//            a workgroup's turn 96 KB (B = 4096)
//   copy     one array of 6 n floats, in-place scale (the upper bound: two streams)
// hipcc --offload-arch=gfx950 -O3 scripts/ablate_layout.hip -o /tmp/ablate_layout && /tmp/ablate_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TH = 1024;

__device__ __forceinline__ void work(f4 (&v)[6])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[0][q] = __builtin_fmaf(1e-3f, v[3][q], v[0][q]); v[1][q] = __builtin_fmaf(1e-3f, v[4][q], v[1][q]); v[2][q] = __builtin_fmaf(1e-3f, v[5][q], v[2][q]);
        v[3][q] *= 0.999f; v[4][q] *= 0.999f; v[5][q] *= 0.999f;
    }
}

__global__ __launch_bounds__(TH) void soa_kernel(float* slab, size_t stride, size_t chunk, size_t n)
{
    const size_t b = blockIdx.x * chunk, e = b + chunk < n ? b + chunk : n;
    for (size_t i = b + threadIdx.x * 4; i < e; i += TH * 4) {
        f4 v[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(slab + f * stride + i));
        work(v);
#pragma unroll
        for (int f = 0; f < 6; ++f) __builtin_nontemporal_store(v[f], reinterpret_cast<f4*>(slab + f * stride + i));
    }
}

template <int B>
__global__ __launch_bounds__(TH) void aosoa_kernel(float* slab, size_t chunk, size_t n)
{
    const size_t b = blockIdx.x * chunk, e = b + chunk < n ? b + chunk : n;
    for (size_t i = b + threadIdx.x * 4; i < e; i += TH * 4) {
        float* p = slab + (i / B) * (6 * B) + (i % B);
        f4 v[6];
#pragma unroll
        for (int f = 0; f < 6; ++f) v[f] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + f * B));
        work(v);
#pragma unroll
        for (int f = 0; f < 6; ++f) __builtin_nontemporal_store(v[f], reinterpret_cast<f4*>(p + f * B));
    }
}

__global__ __launch_bounds__(TH) void copy_kernel(float* a, size_t chunk6, size_t n6)
{
    const size_t b = blockIdx.x * chunk6, e = b + chunk6 < n6 ? b + chunk6 : n6;
    for (size_t i = b + threadIdx.x * 4; i < e; i += TH * 4) {
        f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + i));
        v *= 0.999f;
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(a + i));
    }
}

int main()
{
    const size_t n = 500000000ull / 65536 * 65536, chunk = 65536;
    float* slab;
    CK(hipMalloc(&slab, 6 * n * sizeof(float)));
    CK(hipMemset(slab, 0, 6 * n * sizeof(float)));
    const unsigned grid = static_cast<unsigned>(n / chunk);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-28s best %.3f ms  mean %.3f ms  %.2f TB/s (48 B x %.1e particles)\n", name, best, sum / 5, 48.0 * n / (best * 1e-3) / 1e12, double(n));
    };
    time("soa, 65536 per workgroup", [&] { soa_kernel<<<grid, TH>>>(slab, n, chunk, n); });
    time("aosoa B = 256", [&] { aosoa_kernel<256><<<grid, TH>>>(slab, chunk, n); });
    time("aosoa B = 1024", [&] { aosoa_kernel<1024><<<grid, TH>>>(slab, chunk, n); });
    time("aosoa B = 4096", [&] { aosoa_kernel<4096><<<grid, TH>>>(slab, chunk, n); });
    time("aosoa B = 65536", [&] { aosoa_kernel<65536><<<grid, TH>>>(slab, chunk, n); });
    time("copy (one array of 6 n)", [&] { copy_kernel<<<grid, TH>>>(slab, 6 * chunk, 6 * n); });
    // the same with 256-thread workgroups of 16384 particles (four per CU)
    return 0;
}
