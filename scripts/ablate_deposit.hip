// ablate_deposit.hip — development probe (not product): what bounds the LDS-tiled scatter?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics \
//         -I fusion-sim_amd/csrc scripts/ablate_deposit.hip -o build_probe/ablate_deposit
#include "fpic_kernels.hpp"

#include <cstdio>
#include <cstdlib>

using namespace fpic;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { NO_ATOMICS = 1, PRIVATE_ADDR = 2, FAST_MATH = 4, NO_FLUSH = 8, ONE_CHANNEL = 16, INT_ATOMICS = 32, NO_LOAD_VEL = 64 };

template <int M, int BS, int CHUNK>
__global__ __launch_bounds__(BS) void sums_v(ParticleArrays<float> p, size_t n, int nr, int nz, int ntx, size_t per_tile,
                                             float* __restrict__ cell_sums)
{
    constexpr int PPT = 4;
    constexpr int LW = kTileLds;
    __shared__ float tile[LW * LW * 4];
    // synthetic work list: tile t owns particles [t*per_tile, (t+1)*per_tile), cut into CHUNK pieces
    const size_t chunks_per_tile = (per_tile + CHUNK - 1) / CHUNK;
    const size_t t = blockIdx.x / chunks_per_tile, c = blockIdx.x % chunks_per_tile;
    const size_t begin = t * per_tile + c * CHUNK;
    size_t end = begin + CHUNK;
    if (end > (t + 1) * per_tile) end = (t + 1) * per_tile;
    if (end > n) end = n;
    if (begin >= end) return;
    const int i0 = static_cast<int>(t % ntx) * kTileSide - kTileHalo;
    const int j0 = static_cast<int>(t / ntx) * kTileSide - kTileHalo;
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) tile[k] = 0.f;
    __syncthreads();
    const size_t gw = static_cast<size_t>(nr) + 1;
    float sink = 0.f;
    const size_t first = (begin / PPT) * PPT;
    for (size_t base = first + static_cast<size_t>(threadIdx.x) * PPT; base < end; base += BS * PPT) {
        float x[PPT], y[PPT], z[PPT], vx[PPT] = {}, vy[PPT] = {}, vz[PPT] = {};
        load_lane<float, PPT>(p.x, base, x); load_lane<float, PPT>(p.y, base, y); load_lane<float, PPT>(p.z, base, z);
        if (!(M & NO_LOAD_VEL)) {
            load_lane<float, PPT>(p.vx, base, vx); load_lane<float, PPT>(p.vy, base, vy); load_lane<float, PPT>(p.vz, base, vz);
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const size_t i = base + k;
            if (i < begin || i >= end) continue;
            float r, dx, dy;
            if (M & FAST_MATH) {
                const float q = x[k] * x[k] + y[k] * y[k];
                const float ir = __frsqrt_rn(q);
                r = q * ir; dx = x[k] * ir; dy = y[k] * ir;
            } else {
                r = sqrtf(x[k] * x[k] + y[k] * y[k]); dx = x[k] / r; dy = y[k] / r;
            }
            if (!(r >= 0.f && r <= 1.f && z[k] >= 0.f && z[k] <= 1.f)) continue;
            const int ic = static_cast<int>(r * nr), jc = static_cast<int>(z[k] * nz);
            const float c0 = 0.001f * (vx[k] * dx + vy[k] * dy), c1 = 0.001f * (vy[k] * dx - vx[k] * dy);
            const float c2 = 0.001f * vz[k], c3 = 0.001f;
            const int li = ic - i0, lj = jc - j0;
            if (M & NO_ATOMICS) { sink += c0 + c1 + c2 + c3 + li + lj; continue; }
            if (li >= 0 && li < LW && lj >= 0 && lj < LW) {
                float* t4 = (M & PRIVATE_ADDR) ? tile + 4 * ((threadIdx.x * 9 + k) % (LW * LW)) : tile + 4 * (lj * LW + li);
                if (M & INT_ATOMICS) {
                    atomicAdd(reinterpret_cast<unsigned*>(t4), 1u);
                    if (!(M & ONE_CHANNEL)) {
                        atomicAdd(reinterpret_cast<unsigned*>(t4 + 1), 2u); atomicAdd(reinterpret_cast<unsigned*>(t4 + 2), 3u);
                        atomicAdd(reinterpret_cast<unsigned*>(t4 + 3), 4u);
                    }
                } else {
                    atomicAdd(t4, c0);
                    if (!(M & ONE_CHANNEL)) { atomicAdd(t4 + 1, c1); atomicAdd(t4 + 2, c2); atomicAdd(t4 + 3, c3); }
                }
            } else {
                float* g = cell_sums + 4 * (static_cast<size_t>(ic) + gw * jc);
                atomicAdd(g, c0); atomicAdd(g + 1, c1); atomicAdd(g + 2, c2); atomicAdd(g + 3, c3);
            }
        }
    }
    if (M & NO_ATOMICS) { if (sink == 123.456f) cell_sums[0] = sink; }
    __syncthreads();
    if (M & NO_FLUSH) { if (tile[threadIdx.x] == 123.456f) cell_sums[1] = 1.f; return; }
    for (int k = threadIdx.x; k < LW * LW * 4; k += BS) {
        const float v = tile[k];
        if (v == 0.f) continue;
        const int lj = k / (LW * 4);
        const int rem = k - lj * (LW * 4);
        const int gi = i0 + (rem >> 2), gj = j0 + lj;
        if (gi < 0 || gi > nr || gj < 0 || gj > nz) continue;
        atomicAdd(cell_sums + 4 * (static_cast<size_t>(gi) + gw * gj) + (rem & 3), v);
    }
}

__global__ void init_k(ParticleArrays<float> p, size_t n, int grid, size_t per_tile, float drift)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto rnd = [&](unsigned long long k) {
        unsigned long long z = (i * 16 + k) * 0x9E3779B97F4A7C15ull + 0x1234567;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
    };
    const int nt = grid / 32;
    const size_t t = i / per_tile;
    float rh = ((t % nt) * 32 + rnd(0) * 32 + drift * (rnd(10) - 0.5f)) / grid;
    float zh = ((t / nt) * 32 + rnd(1) * 32 + drift * (rnd(11) - 0.5f)) / grid;
    rh = fminf(fmaxf(rh, 0.001f), 0.999f); zh = fminf(fmaxf(zh, 0.001f), 0.999f);
    const float th = 6.2831853f * rnd(2);
    p.x[i] = rh * cosf(th); p.y[i] = rh * sinf(th); p.z[i] = zh;
    p.vx[i] = 1e-3f * (rnd(3) - 0.5f); p.vy[i] = 1e-3f * (rnd(4) - 0.5f); p.vz[i] = 1e-3f * (rnd(5) - 0.5f);
}

template <int M, int BS, int CHUNK>
float run(ParticleArrays<float> p, size_t n, int grid, size_t per_tile, float* sums, int reps)
{
    const int ntx = grid / 32;
    const size_t ntiles = static_cast<size_t>(ntx) * ntx;
    const size_t chunks_per_tile = (per_tile + CHUNK - 1) / CHUNK;
    const unsigned nb = static_cast<unsigned>(ntiles * chunks_per_tile);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    sums_v<M, BS, CHUNK><<<nb, BS>>>(p, n, grid, grid, ntx, per_tile, sums);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) sums_v<M, BS, CHUNK><<<nb, BS>>>(p, n, grid, grid, ntx, per_tile, sums);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t n = 100000000ull;
    const int grid = 1024;
    const size_t per_tile = n / 1024 + 1;
    ParticleArrays<float> p{};
    float** arrs[] = { &p.x, &p.y, &p.z, &p.vx, &p.vy, &p.vz };
    for (auto a : arrs) CK(hipMalloc(reinterpret_cast<void**>(a), (n + 4096) * sizeof(float)));
    float* sums;
    CK(hipMalloc(reinterpret_cast<void**>(&sums), 1025ull * 1025 * 16));
    CK(hipMemset(sums, 0, 1025ull * 1025 * 16));
    for (float drift : { 0.f, 8.f }) {
        init_k<<<(n + 255) / 256, 256>>>(p, n, grid, per_tile, drift);
        CK(hipDeviceSynchronize());
        printf("---- drift +-%.0f cells, chunk 32768, block 256\n", drift / 2);
        printf("full                         %.3f ms\n", run<0, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("no LDS atomics               %.3f ms\n", run<NO_ATOMICS, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("no atomics + fast math       %.3f ms\n", run<NO_ATOMICS | FAST_MATH, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("no atomics, no vel loads     %.3f ms\n", run<NO_ATOMICS | NO_LOAD_VEL, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("lane-private addresses       %.3f ms\n", run<PRIVATE_ADDR, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("one channel only             %.3f ms\n", run<ONE_CHANNEL, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("integer atomics              %.3f ms\n", run<INT_ATOMICS, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("no flush                     %.3f ms\n", run<NO_FLUSH, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("fast math                    %.3f ms\n", run<FAST_MATH, 256, 32768>(p, n, grid, per_tile, sums, 5));
        printf("full, block 512              %.3f ms\n", run<0, 512, 32768>(p, n, grid, per_tile, sums, 5));
        printf("full, block 1024             %.3f ms\n", run<0, 1024, 32768>(p, n, grid, per_tile, sums, 5));
        printf("full, chunk 98304            %.3f ms\n", run<0, 256, 98304>(p, n, grid, per_tile, sums, 5));
        printf("full, chunk 8192             %.3f ms\n", run<0, 256, 8192>(p, n, grid, per_tile, sums, 5));
    }
    return 0;
}
