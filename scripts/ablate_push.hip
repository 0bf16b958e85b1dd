// ablate_push.hip — development probe (not product, not shipped): times variants of the
// push kernel with pieces removed, to find what bounds it.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics \
//         -I fusion-sim_amd/csrc scripts/ablate_push.hip -o /tmp/ablate_push && /tmp/ablate_push
#include "fpic_kernels.hpp"
#include "fpic_push.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fpic;

struct ProbeArgs {
    ParticleArrays<float> p;
    const float* coef; const uint8_t* sink_alive; const float* inv_cdf_xy; const float* entropy;
    int nr, nz; float step_factor; unsigned long long n; int nsub;
};

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { NO_ENTROPY = 1, NO_COEF = 2, NO_SINK = 4, FAST_MATH = 8, NO_STORE = 16, NO_RAND_IO = 32, ENTROPY_1MB = 64, ENTROPY_16KB = 128, ENTROPY_4MB = 256,
       HOT_FIRST = 512, HOT_ROWS = 1024, // timing only: the texel's place under a layout that puts the hot texels first (round 4)
       GATHER_AUX_SHIFT = 12 }; // bits 12..: (aux + 1) of a buffer-instruction gather: 1 = sc0, 2 = nt, 16 = sc1, sums thereof

typedef unsigned int nat_b128 __attribute__((ext_vector_type(4)));

// Round 4: where would texel (i, j) of the entropy table sit if the table were laid out hot texels first?  The logistic
// map's invariant density is the arcsine law, so lookups concentrate where i or j is near 0 or 1023: heat ~ 1/sqrt(a b)
// with a, b the distances to the nearer edge.  Levels la = bit length of a (0..9), blocks (la, lb) of 2^(la-1) x 2^(lb-1) x 4
// texels in the order of la + lb.  g_hot_base[la][lb] = first texel of the block.
__constant__ int g_hot_base[10][10];
__device__ __forceinline__ int hot_first_index(int i, int j)
{
    const int a = min(i, 1023 - i), b = min(j, 1023 - j);
    const int la = 32 - __clz(a), lb = 32 - __clz(b);                 // 0 for a == 0, else floor(log2 a) + 1
    const int wa = la ? 1 << (la - 1) : 1, wb = lb ? 1 << (lb - 1) : 1; // block extents
    const int oa = la ? a - wa : 0, ob = lb ? b - wb : 0;
    const int quad = (i > 511) | ((j > 511) << 1);
    return g_hot_base[la][lb] + ((ob * wa + oa) << 2) + quad;
}
static void fill_hot_base()
{
    int base[10][10], next = 0;
    for (int g = 0; g <= 18; ++g)
        for (int la = 0; la < 10; ++la) {
            const int lb = g - la;
            if (lb < 0 || lb > 9) continue;
            base[la][lb] = next;
            next += 4 * (la ? 1 << (la - 1) : 1) * (lb ? 1 << (lb - 1) : 1);
        }
    if (next != 1024 * 1024) { printf("hot layout covers %d texels\n", next); exit(1); }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hot_base), base, sizeof base));
}

template <int M>
__device__ __forceinline__ void substep_v(Particle<float>& q, const ProbeArgs& a)
{
    float s[4] = { 0.3f, 0.7f, 0.2f, 0.9f };
    if (!(M & NO_ENTROPY)) {
        int et = ngp(q.c1, kEntropySide) + kEntropySide * ngp(q.c2, kEntropySide);
        if (M & ENTROPY_1MB) et &= 0xFFFF;      // timing only: 64K texels = 1 MB footprint
        if (M & ENTROPY_4MB) et &= 0x3FFFF;     // 4 MB
        if (M & ENTROPY_16KB) et &= 0x3FF;      // 16 KB
        if (M & HOT_FIRST) et = hot_first_index(ngp(q.c1, kEntropySide), ngp(q.c2, kEntropySide));
        if (M & HOT_ROWS) { const int j = ngp(q.c2, kEntropySide), b = min(j, 1023 - j); et = ngp(q.c1, kEntropySide) + kEntropySide * (2 * b + (j > 511)); } // rows folded: hot rows first
        constexpr int aux1 = M >> GATHER_AUX_SHIFT;
        if constexpr (aux1 == 0) {
            load4(a.entropy + 4 * static_cast<size_t>(et), s);
        } else {
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.entropy), 0, 0xFFFFFFF0u, 0x00020000);
            const nat_b128 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, et * 16, 0, aux1 - 1);
            s[0] = __uint_as_float(raw.x); s[1] = __uint_as_float(raw.y); s[2] = __uint_as_float(raw.z); s[3] = __uint_as_float(raw.w);
        }
    }
    float r, dx, dy;
    if (M & FAST_MATH) {
        const float ir = __frsqrt_rn(q.x * q.x + q.y * q.y);
        r = (q.x * q.x + q.y * q.y) * ir; dx = q.x * ir; dy = q.y * ir;
    } else {
        r = sqrtf(q.x * q.x + q.y * q.y); dx = q.x / r; dy = q.y / r;
    }
    const float vr = q.vx * dx + q.vy * dy;
    const float va = q.vy * dx - q.vx * dy;
    const int cell = ngp(r, a.nr) + a.nr * ngp(q.z, a.nz);
    float R1[4] = { 1, 0.01f, 0, 0 }, R2[4] = { -0.01f, 1, 0, 0 }, R3[4] = { 0, 0, 1, 0 };
    if (!(M & NO_COEF)) {
        const float* cf = a.coef + 12 * static_cast<size_t>(cell);
        load4(cf, R1); load4(cf + 4, R2); load4(cf + 8, R3);
    }
    const float cx = ((R1[0] * vr + R1[1] * va) + R1[2] * q.vz) + R1[3];
    const float cy = ((R2[0] * vr + R2[1] * va) + R2[2] * q.vz) + R2[3];
    const float cz = ((R3[0] * vr + R3[1] * va) + R3[2] * q.vz) + R3[3];
    float nvx = cx * dx - cy * dy, nvy = cx * dy + cy * dx, nvz = cz;
    if (!q.alive) { nvx = 0.001f * (2.f * q.u1 - 1.f); nvy = 0.001f * (2.f * q.u2 - 1.f); nvz = 0.001f * (2.f * q.c1 - 1.f); }
    const float nx = q.x + a.step_factor * nvx, ny = q.y + a.step_factor * nvy, nzp = q.z + a.step_factor * nvz;
    float r2;
    if (M & FAST_MATH) { const float t = nx * nx + ny * ny; r2 = t * __frsqrt_rn(t); }
    else r2 = sqrtf(nx * nx + ny * ny);
    const int cell2 = ngp(r2, a.nr) + a.nr * ngp(nzp, a.nz);
    bool keep = true;
    if (!(M & NO_SINK)) keep = a.sink_alive[cell2] != 0;
    if (keep) { q.x = nx; q.y = ny; q.z = nzp; }
    else {
        const int t = ngp(q.u1, kCdfSide) + kCdfSide * ngp(q.u2, kCdfSide);
        q.x = a.inv_cdf_xy[2 * static_cast<size_t>(t)]; q.y = 0.f; q.z = a.inv_cdf_xy[2 * static_cast<size_t>(t) + 1];
    }
    q.alive = keep; q.vx = nvx; q.vy = nvy; q.vz = nvz;
    const float x0 = 0.999f * q.c1 + 0.001f * s[2], x1 = 0.999f * q.c2 + 0.001f * s[3];
    const float m0 = q.u1 + s[0], m1 = q.u2 + s[1];
    q.u1 = (m0 > 1.f) ? m0 - 1.f : m0; q.u2 = (m1 > 1.f) ? m1 - 1.f : m1;
    q.c1 = 4.f * x0 * (1.f - x0); q.c2 = 4.f * x1 * (1.f - x1);
}

template <int M, int BS>
__global__ __launch_bounds__(BS) void push_v(ProbeArgs a)
{
    constexpr int PPT = 4;
    const size_t base = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * PPT;
    if (base >= a.n) return;
    float x[PPT], y[PPT], z[PPT], vx[PPT], vy[PPT], vz[PPT], u1[PPT], u2[PPT], c1[PPT], c2[PPT];
    load_lane<float, PPT>(a.p.x, base, x); load_lane<float, PPT>(a.p.y, base, y); load_lane<float, PPT>(a.p.z, base, z);
    load_lane<float, PPT>(a.p.vx, base, vx); load_lane<float, PPT>(a.p.vy, base, vy); load_lane<float, PPT>(a.p.vz, base, vz);
    if (!(M & NO_RAND_IO)) {
        load_lane<float, PPT>(a.p.u1, base, u1); load_lane<float, PPT>(a.p.u2, base, u2);
        load_lane<float, PPT>(a.p.c1, base, c1); load_lane<float, PPT>(a.p.c2, base, c2);
    } else {
        for (int k = 0; k < PPT; ++k) { u1[k] = 0.1f + 0.2f * k; u2[k] = 0.3f; c1[k] = x[k] * 0.9f; c2[k] = z[k] * 0.9f; }
    }
    const uchar4 av = *reinterpret_cast<const uchar4*>(a.p.alive + base);
    const uint8_t al[4] = { av.x, av.y, av.z, av.w };
    Particle<float> q[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        q[k].x = x[k]; q[k].y = y[k]; q[k].z = z[k]; q[k].vx = vx[k]; q[k].vy = vy[k]; q[k].vz = vz[k];
        q[k].u1 = u1[k]; q[k].u2 = u2[k]; q[k].c1 = c1[k]; q[k].c2 = c2[k]; q[k].alive = al[k] != 0;
    }
    for (int s = 0; s < a.nsub; ++s) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) substep_v<M>(q[k], a);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        x[k] = q[k].x; y[k] = q[k].y; z[k] = q[k].z; vx[k] = q[k].vx; vy[k] = q[k].vy; vz[k] = q[k].vz;
        u1[k] = q[k].u1; u2[k] = q[k].u2; c1[k] = q[k].c1; c2[k] = q[k].c2;
    }
    if (M & NO_STORE) {
        float acc = 0;
        for (int k = 0; k < PPT; ++k) acc += x[k] + y[k] + z[k] + vx[k] + vy[k] + vz[k] + u1[k] + u2[k] + c1[k] + c2[k];
        if (acc == 123.456f) a.p.x[base] = acc;
        return;
    }
    store_lane<float, PPT>(a.p.x, base, x); store_lane<float, PPT>(a.p.y, base, y); store_lane<float, PPT>(a.p.z, base, z);
    store_lane<float, PPT>(a.p.vx, base, vx); store_lane<float, PPT>(a.p.vy, base, vy); store_lane<float, PPT>(a.p.vz, base, vz);
    if (!(M & NO_RAND_IO)) {
        store_lane<float, PPT>(a.p.u1, base, u1); store_lane<float, PPT>(a.p.u2, base, u2);
        store_lane<float, PPT>(a.p.c1, base, c1); store_lane<float, PPT>(a.p.c2, base, c2);
    }
    *reinterpret_cast<uchar4*>(a.p.alive + base) = make_uchar4(q[0].alive, q[1].alive, q[2].alive, q[3].alive);
}

__global__ void init_k(ParticleArrays<float> p, size_t n, int sorted, int grid);

// Round 4: K3 alone.  The random state's advance (empic.js:783-820) reads nothing of the particle's motion: what if it ran as
// a kernel of its own — 16 B in, the two dependent table gathers, 16 B out — beside a push that streams 50 B?
template <int BS, int NT>
__global__ __launch_bounds__(BS) void rng_only(ProbeArgs a)
{
    constexpr int PPT = 4;
    const size_t base = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * PPT;
    if (base >= a.n) return;
    float u1[PPT], u2[PPT], c1[PPT], c2[PPT];
    load_lane<float, PPT>(a.p.u1, base, u1); load_lane<float, PPT>(a.p.u2, base, u2);
    load_lane<float, PPT>(a.p.c1, base, c1); load_lane<float, PPT>(a.p.c2, base, c2);
    for (int s = 0; s < a.nsub; ++s) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            float e[4];
            const int et = ngp(c1[k], kEntropySide) + kEntropySide * ngp(c2[k], kEntropySide);
            load4(a.entropy + 4 * static_cast<size_t>(et), e);
            const float x0 = 0.999f * c1[k] + 0.001f * e[2], x1 = 0.999f * c2[k] + 0.001f * e[3];
            const float m0 = u1[k] + e[0], m1 = u2[k] + e[1];
            u1[k] = (m0 > 1.f) ? m0 - 1.f : m0; u2[k] = (m1 > 1.f) ? m1 - 1.f : m1;
            c1[k] = 4.f * x0 * (1.f - x0); c2[k] = 4.f * x1 * (1.f - x1);
        }
    }
    store_lane<float, PPT>(a.p.u1, base, u1); store_lane<float, PPT>(a.p.u2, base, u2);
    store_lane<float, PPT>(a.p.c1, base, c1); store_lane<float, PPT>(a.p.c2, base, c2);
}

template <int BS>
float run_rng(ProbeArgs a, int reps, ParticleArrays<float> p, int grid)
{
    init_k<<<(a.n + 255) / 256, 256>>>(p, a.n, 1, grid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = static_cast<unsigned>((a.n / 4 + BS - 1) / BS);
    rng_only<BS, 0><<<nb, BS>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) rng_only<BS, 0><<<nb, BS>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

__global__ void init_k(ParticleArrays<float> p, size_t n, int sorted, int grid)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto rnd = [&](unsigned long long k) {
        unsigned long long z = (i * 16 + k) * 0x9E3779B97F4A7C15ull + 0x1234567;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        return static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
    };
    float rh, zh;
    if (sorted) {
        const int nt = grid / 32;
        const size_t per = n / (static_cast<size_t>(nt) * nt) + 1;
        const size_t t = i / per;
        rh = ((t % nt) * 32 + rnd(0) * 32) / grid; zh = ((t / nt) * 32 + rnd(1) * 32) / grid;
        rh = fminf(fmaxf(rh, 0.01f), 0.98f); zh = fminf(fmaxf(zh, 0.01f), 0.98f);
    } else { rh = 0.01f + 0.97f * sqrtf(rnd(0)); zh = 0.01f + 0.97f * rnd(1); }
    const float th = 6.2831853f * rnd(2);
    p.x[i] = rh * cosf(th); p.y[i] = rh * sinf(th); p.z[i] = zh;
    p.vx[i] = 1e-3f * (rnd(3) - 0.5f); p.vy[i] = 1e-3f * (rnd(4) - 0.5f); p.vz[i] = 1e-3f * (rnd(5) - 0.5f);
    p.u1[i] = rnd(6); p.u2[i] = rnd(7); p.c1[i] = rnd(8); p.c2[i] = rnd(9);
    p.alive[i] = 1;
}

__global__ void fill_k(float* p, size_t n, float lo, float hi)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (i + 77) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
    p[i] = lo + (hi - lo) * static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
}

template <int M, int BS>
float run(ProbeArgs a, int reps, ParticleArrays<float> p, int sorted, int grid)
{
    init_k<<<(a.n + 255) / 256, 256>>>(p, a.n, sorted, grid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = static_cast<unsigned>((a.n / 4 + BS - 1) / BS);
    push_v<M, BS><<<nb, BS>>>(a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) push_v<M, BS><<<nb, BS>>>(a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
    const int grid = 1024;
    ParticleArrays<float> p;
    float** arrs[] = { &p.x, &p.y, &p.z, &p.vx, &p.vy, &p.vz, &p.u1, &p.u2, &p.c1, &p.c2 };
    for (auto a : arrs) CK(hipMalloc(reinterpret_cast<void**>(a), (n + 1024) * sizeof(float)));
    CK(hipMalloc(reinterpret_cast<void**>(&p.alive), n + 1024));
    p.id = nullptr;
    float *coef, *inv, *ent; uint8_t* sink;
    const size_t nc = static_cast<size_t>(grid) * grid;
    CK(hipMalloc(reinterpret_cast<void**>(&coef), nc * 12 * 4)); CK(hipMalloc(reinterpret_cast<void**>(&inv), 512 * 512 * 2 * 4));
    CK(hipMalloc(reinterpret_cast<void**>(&ent), 1024 * 1024 * 4 * 4)); CK(hipMalloc(reinterpret_cast<void**>(&sink), nc));
    fill_k<<<(nc * 12 + 255) / 256, 256>>>(coef, nc * 12, -0.01f, 0.01f);
    fill_k<<<(512 * 512 * 2 + 255) / 256, 256>>>(inv, 512 * 512 * 2, 0.1f, 0.9f);
    fill_k<<<(1024 * 1024 * 4 + 255) / 256, 256>>>(ent, 1024 * 1024 * 4, 0.f, 1.f);
    CK(hipMemset(sink, 1, nc));
    ProbeArgs a;
    a.p = p; a.coef = coef; a.sink_alive = sink; a.inv_cdf_xy = inv; a.entropy = ent;
    a.nr = grid; a.nz = grid; a.step_factor = 0.5996f; a.n = n; a.nsub = 2;
    const int reps = 5;
    fill_hot_base();
    printf("---- K3 alone (random state: 16 B in, 2 dependent gathers, 16 B out), n=%zu\n", n);
    printf("rng only, block 256        %.3f ms\n", run_rng<256>(a, reps, p, grid));
    printf("rng only, block 1024       %.3f ms\n", run_rng<1024>(a, reps, p, grid));
    printf("push without the random state: no entropy, no rand I/O, no coef/sink gathers  %.3f ms\n", run<NO_ENTROPY | NO_RAND_IO | NO_COEF | NO_SINK, 256>(a, reps, p, 1, grid));
    printf("push without the random state, with coef/sink gathers                         %.3f ms\n", run<NO_ENTROPY | NO_RAND_IO, 256>(a, reps, p, 1, grid));
    for (int sorted = 1; sorted >= 0; --sorted) {
        printf("---- particles %s, n=%zu, nsub=2\n", sorted ? "tile-sorted" : "random order", n);
        printf("full                       %.3f ms\n", run<0, 256>(a, reps, p, sorted, grid));
        printf("no entropy gather          %.3f ms\n", run<NO_ENTROPY, 256>(a, reps, p, sorted, grid));
        printf("no coef gather             %.3f ms\n", run<NO_COEF, 256>(a, reps, p, sorted, grid));
        printf("no sink gather             %.3f ms\n", run<NO_SINK, 256>(a, reps, p, sorted, grid));
        printf("no gathers at all          %.3f ms\n", run<NO_ENTROPY | NO_COEF | NO_SINK, 256>(a, reps, p, sorted, grid));
        printf("fast math (rsq, no div)    %.3f ms\n", run<FAST_MATH, 256>(a, reps, p, sorted, grid));
        printf("no gathers + fast math     %.3f ms\n", run<NO_ENTROPY | NO_COEF | NO_SINK | FAST_MATH, 256>(a, reps, p, sorted, grid));
        printf("no stores                  %.3f ms\n", run<NO_STORE, 256>(a, reps, p, sorted, grid));
        printf("no rand I/O (6 streams)    %.3f ms\n", run<NO_RAND_IO, 256>(a, reps, p, sorted, grid));
        printf("no coef/sink (~LDS-staged) %.3f ms\n", run<NO_COEF | NO_SINK, 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load aux 0        %.3f ms\n", run<NO_COEF | NO_SINK | (1 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load sc0          %.3f ms\n", run<NO_COEF | NO_SINK | (2 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load nt           %.3f ms\n", run<NO_COEF | NO_SINK | (3 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load sc0 nt       %.3f ms\n", run<NO_COEF | NO_SINK | (4 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load sc1          %.3f ms\n", run<NO_COEF | NO_SINK | (17 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load sc0 sc1      %.3f ms\n", run<NO_COEF | NO_SINK | (18 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, buffer load sc1 nt       %.3f ms\n", run<NO_COEF | NO_SINK | (19 << GATHER_AUX_SHIFT), 256>(a, reps, p, sorted, grid));
        printf(" + entropy, hot texels first %.3f ms\n", run<NO_COEF | NO_SINK | HOT_FIRST, 256>(a, reps, p, sorted, grid));
        printf(" + entropy, hot rows first   %.3f ms\n", run<NO_COEF | NO_SINK | HOT_ROWS, 256>(a, reps, p, sorted, grid));
        printf(" + entropy footprint 4 MB  %.3f ms\n", run<NO_COEF | NO_SINK | ENTROPY_4MB, 256>(a, reps, p, sorted, grid));
        printf(" + entropy footprint 1 MB  %.3f ms\n", run<NO_COEF | NO_SINK | ENTROPY_1MB, 256>(a, reps, p, sorted, grid));
        printf(" + entropy footprint 16 KB %.3f ms\n", run<NO_COEF | NO_SINK | ENTROPY_16KB, 256>(a, reps, p, sorted, grid));
        printf("full, block 128            %.3f ms\n", run<0, 128>(a, reps, p, sorted, grid));
        printf("full, block 512            %.3f ms\n", run<0, 512>(a, reps, p, sorted, grid));
    }
    a.nsub = 8;
    printf("---- nsub=8 (step(4)), tile-sorted\n");
    printf("full                       %.3f ms\n", run<0, 256>(a, reps, p, 1, grid));
    printf("no entropy                 %.3f ms\n", run<NO_ENTROPY, 256>(a, reps, p, 1, grid));
    printf("no gathers + fast math     %.3f ms\n", run<NO_ENTROPY | NO_COEF | NO_SINK | FAST_MATH, 256>(a, reps, p, 1, grid));
    return 0;
}
