// ablate_push3.hip — development probe (not part of the library): where the CART3D push kernel's time goes.
// Builds a uniform plasma already sorted by tile (256^3 nodes, P particles per cell), a random field, and
// times push3_tiles_kernel<float> with parts switched off and with 512 / 1024 threads per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I fusion-sim_amd/csrc -I include scripts/ablate_push3.hip -o /tmp/ablate_push3
#include <hip/hip_runtime.h>
#include "fpic_handle.hpp"
#include "fes_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace fes;

__device__ unsigned long long mix(unsigned long long z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__device__ float u01(unsigned long long z) { return (mix(z) >> 40) * (1.0f / 16777216.0f); }

// particle p of tile t: uniform inside the tile; sorted=1: ordered by cell inside the tile as well
__global__ void fill(float* slab, size_t stride, size_t n, int per_tile, int g, int ntx, int nty, int sorted) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (p >= n) return;
    int t = (int)(p / per_tile), q = (int)(p % per_tile);
    int ti = t % ntx, tj = (t / ntx) % nty, tk = t / (ntx * nty);
    float fx = u01(3 * p + 1), fy = u01(3 * p + 2), fz = u01(3 * p + 3);
    float cx, cy, cz;
    if (sorted) { int cells = kTX * kTY * kTZ; int c = (int)((long long)q * cells / per_tile); cx = c % kTX + fx; cy = (c / kTX) % kTY + fy; cz = c / (kTX * kTY) + fz; }
    else { cx = fx * kTX; cy = fy * kTY; cz = fz * kTZ; }
    slab[p] = (ti * kTX + cx) / g; slab[stride + p] = (tj * kTY + cy) / g; slab[2 * stride + p] = (tk * kTZ + cz) / g;
    for (int f = 3; f < 6; ++f) slab[f * stride + p] = 1e-3f * (u01(7 * p + f) - 0.5f);
}
__global__ void fillE(float* E4, size_t n4) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n4) E4[i] = 1e3f * (u01(i + 99) - 0.5f); }
__global__ void fillwork(BlockWork* w, uint32_t* nwork, int ntiles, int per_tile, int chunk) {
    int t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= ntiles) return;
    int per = (per_tile + chunk - 1) / chunk;
    for (int b = 0; b < per; ++b) { BlockWork x; x.tile = t; x.begin = (uint32_t)((size_t)t * per_tile + (size_t)b * chunk); uint32_t e = x.begin + chunk; uint32_t lim = (uint32_t)((size_t)(t + 1) * per_tile); x.end = e < lim ? e : lim; x.pad = 0; w[t * per + b] = x; }
    if (t == 0) *nwork = ntiles * per;
}

template <int THREADS, int ABL>
float run(Push3Args<float> one, unsigned grid, int reps) {
    const size_t lds = push3_lds_bytes<float>();
    Push3Joint<float> a{};   // (one species with its own work list: chunk = 0)
    a.nsp = 1; a.sp[0] = one; a.work = one.work; a.nwork = one.nwork; a.chunk = 0;
    CK(hipFuncSetAttribute((const void*)push3_tiles_kernel<float, false, false, false, THREADS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    push3_tiles_kernel<float, false, false, false, THREADS, ABL><<<grid, THREADS, lds>>>(a); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) { CK(hipMemsetAsync(one.rho, 0, (size_t)one.nx * one.ny * one.nz * 8)); push3_tiles_kernel<float, false, false, false, THREADS, ABL><<<grid, THREADS, lds>>>(a); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 256, ppc = argc > 2 ? atoi(argv[2]) : 30, sorted = argc > 3 ? atoi(argv[3]) : 0;
    const int chunk = argc > 4 ? atoi(argv[4]) : kChunk3;
    const int ntx = g / kTX, nty = g / kTY, ntz = g / kTZ, ntiles = ntx * nty * ntz;
    const int per_tile = ppc * kTX * kTY * kTZ;
    const size_t n = (size_t)ntiles * per_tile, stride = (n + 1023) / 1024 * 1024, nodes = (size_t)g * g * g;
    float *slab, *E4; unsigned long long *rho, *spilled; BlockWork* work; uint32_t* nwork;
    const int per = (per_tile + chunk - 1) / chunk;
    CK(hipMalloc((void**)&slab, 6 * stride * 4)); CK(hipMalloc((void**)&E4, nodes * 16)); CK(hipMalloc((void**)&rho, nodes * 8));
    uint32_t* tcount; CK(hipMalloc((void**)&tcount, 4 * (ntiles + 1))); CK(hipMemset(tcount, 0, 4 * (ntiles + 1)));
    CK(hipMalloc((void**)&spilled, 8)); CK(hipMalloc((void**)&work, sizeof(BlockWork) * ntiles * per)); CK(hipMalloc((void**)&nwork, 4));
    fill<<<(unsigned)((n + 255) / 256), 256>>>(slab, stride, n, per_tile, g, ntx, nty, sorted);
    fillE<<<(unsigned)((nodes * 4 + 255) / 256), 256>>>(E4, nodes * 4);
    fillwork<<<(ntiles + 255) / 256, 256>>>(work, nwork, ntiles, per_tile, chunk);
    CK(hipMemset(spilled, 0, 8)); CK(hipDeviceSynchronize());
    Push3Args<float> a{};
    a.slab = slab; a.stride = stride; a.n = n; a.E4 = E4; a.rho = rho; a.nx = a.ny = a.nz = g;
    a.held = Held{ 0, g };   // (every plane is held: without this no window would be staged nor flushed — the slab-only arrays of round 3)
    a.hc = 1e-9f; a.dx = a.dy = a.dz = 1e-5f;   // tiny steps: the order stays as generated over all repetitions
    a.Z = 1; a.ntx = ntx; a.nty = nty; a.ntz = ntz; a.work = work; a.nwork = nwork; a.spilled = spilled; a.tile_count = tcount;
    const unsigned grid = ntiles * per;
    printf("grid %d^3, %d per cell (%zu particles, %.2f GB streamed per launch), %s inside tiles, chunk %d, %u workgroups\n", g, ppc, n, 48.0 * n / 1e9,
           sorted ? "cell-sorted" : "random", chunk, grid);
    const int R = 5;
#define ROW(TH, ABL, what) { float ms = run<TH, ABL>(a, grid, R); printf("%-52s %4d threads  %7.3f ms  %6.0f GB/s\n", what, TH, ms, 48.0 * n / ms / 1e6); }
    if (argc > 5) { // short form for density sweeps: the 1024-thread rows only
        ROW(1024, 0, "full kernel");
        ROW(1024, 4, "no window staging or flush");
        ROW(1024, 8, "flush by plain stores instead of atomics");
        ROW(1024, 3, "no LDS accumulation, no gather (staging/flush kept)");
        ROW(1024, 7, "neither, no window staging or flush");
    } else {
    ROW(512, 0, "full kernel");
    ROW(1024, 0, "full kernel");
    ROW(512, 1, "no LDS accumulation");
    ROW(1024, 1, "no LDS accumulation");
    ROW(512, 2, "no field gather");
    ROW(1024, 2, "no field gather");
    ROW(512, 3, "neither (stream + arithmetic + staging/flush)");
    ROW(1024, 3, "neither (stream + arithmetic + staging/flush)");
    ROW(512, 7, "neither, no window staging or flush");
    ROW(1024, 7, "neither, no window staging or flush");
    }
    unsigned long long sp; CK(hipMemcpy(&sp, spilled, 8, hipMemcpyDeviceToHost)); printf("(out-of-window deposits over all runs: %llu)\n", sp);
    return 0;
}
