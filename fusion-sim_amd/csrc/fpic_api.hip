// fpic_api.hip — the C ABI of libfusionpic.so (include/fusionpic.h) on top of the
// gfx950 kernels in fpic_kernels.hpp.  One handle = one reference pusher object
// (empic.js:30-1529) on one GPU; all work is enqueued on the handle's HIP stream.
//
// There is no CPU fallback in this file or anywhere in the library: a handle
// cannot be created without a gfx950 device.
#include "fpic_handle.hpp"
#include "fpic_kernels.hpp"
#include "fpic_push.hpp"
#include "fpic_injection.hpp"
#include "fes_api.hpp"
#include "fpic_comm.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace fpic;

namespace fpic {
std::string& create_error()
{
    thread_local std::string text;
    return text;
}
} // namespace fpic

namespace {

template <typename T>
ParticleArrays<T> arrays(fpic_handle* h, int which)
{
    ParticleArrays<T> p;
    void** a = h->part[which];
    p.x = static_cast<T*>(a[0]); p.y = static_cast<T*>(a[1]); p.z = static_cast<T*>(a[2]);
    p.vx = static_cast<T*>(a[3]); p.vy = static_cast<T*>(a[4]); p.vz = static_cast<T*>(a[5]);
    p.u1 = static_cast<T*>(a[6]); p.u2 = static_cast<T*>(a[7]); p.c1 = static_cast<T*>(a[8]); p.c2 = static_cast<T*>(a[9]);
    p.alive = h->alive[which];
    p.id = h->id[which];
    return p;
}

// counter-based RNG mode: the random vector of caller's particle i at sub-step t
__global__ __launch_bounds__(256) void counter_rand_readback_kernel(float* aos4, size_t chunk_begin, size_t chunk_n, uint32_t k0,
                                                                     uint32_t k1, unsigned long long t)
{
    const size_t o = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (o >= chunk_n) return;
    float u[4];
    counter_rand<float>(static_cast<uint32_t>(chunk_begin + o), t, k0, k1, u);
    aos4[4 * o] = u[0]; aos4[4 * o + 1] = u[1]; aos4[4 * o + 2] = u[2]; aos4[4 * o + 3] = u[3];
}

// default random state when the host never injects one: the reference seeds from
// window.crypto / Math.random (empic.js:148-173, quirk Q8)
__global__ __launch_bounds__(256) void default_random_kernel(float* out, size_t n, unsigned long long seed)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
}

template <typename T>
int upload_entropy(fpic_handle* h, const float* dev_f32)
{
    const size_t n = static_cast<size_t>(4) * kEntropySide * kEntropySide;
    convert_kernel<T, float><<<blocks_for(n), 256, 0, h->stream>>>(dev_f32, static_cast<T*>(h->entropy), n);
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

template <typename T>
int upload_rand_chunked(fpic_handle* h, const float* host_rand, const float* dev_all)
{
    // host_rand: [n][4] on the host, staged in chunks; dev_all: [n][4] already on the device
    const size_t chunk = 16u << 20;
    if (dev_all) {
        set_rand_kernel<T><<<blocks_for(h->n), 256, 0, h->stream>>>(dev_all, 0, h->n, arrays<T>(h, h->cur), h->n);
        HIP_TRY(h, hipGetLastError());
        return FPIC_OK;
    }
    float* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, h->n) * 4 * sizeof(float)));
    for (size_t b = 0; b < h->n; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        hipError_t e = hipMemcpyAsync(stage, host_rand + 4 * b, m * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            set_rand_kernel<T><<<blocks_for(h->n), 256, 0, h->stream>>>(stage, b, m, arrays<T>(h, h->cur), h->n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "random-state upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_vec3(fpic_handle* h, const In* host, int first_array, double fxy, double fz, bool set_alive)
{
    const size_t chunk = 8u << 20;
    In* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, h->n) * 3 * sizeof(In)));
    void** a = h->part[h->cur];
    for (size_t b = 0; b < h->n; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            set_vec3_kernel<T, In><<<blocks_for(h->n), 256, 0, h->stream>>>(
                stage, b, m, fxy, fz, static_cast<T*>(a[first_array]), static_cast<T*>(a[first_array + 1]),
                static_cast<T*>(a[first_array + 2]), set_alive ? h->alive[h->cur] : nullptr, h->id[h->cur], h->n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename Out>
int download_vec3(fpic_handle* h, Out* host, int first_array)
{
    const size_t chunk = 8u << 20;
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, h->n) * 3 * sizeof(Out)));
    void** a = h->part[h->cur];
    for (size_t b = 0; b < h->n; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        get_vec3_kernel<T, Out><<<blocks_for(h->n), 256, 0, h->stream>>>(
            static_cast<const T*>(a[first_array]), static_cast<const T*>(a[first_array + 1]),
            static_cast<const T*>(a[first_array + 2]), h->id[h->cur], h->n, b, m, stage);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host + 3 * b, stage, m * 3 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T>
int download_misc(fpic_handle* h, float* rand, uint8_t* alive, int32_t* cells)
{
    const size_t chunk = 16u << 20;
    const size_t m0 = std::min(chunk, h->n);
    float* srand = nullptr;
    uint8_t* salive = nullptr;
    int32_t* scells = nullptr;
    hipError_t ea = hipSuccess;
    if (rand) ea = hipMalloc(reinterpret_cast<void**>(&srand), m0 * 4 * sizeof(float));
    if (ea == hipSuccess && alive) ea = hipMalloc(reinterpret_cast<void**>(&salive), m0);
    if (ea == hipSuccess && cells) ea = hipMalloc(reinterpret_cast<void**>(&scells), m0 * sizeof(int32_t));
    int rc = FPIC_OK;
    if (ea != hipSuccess)
        rc = fail(h, ea == hipErrorOutOfMemory ? FPIC_ERR_OOM : FPIC_ERR_HIP, "read-back staging allocation failed: %s", hipGetErrorString(ea));
    for (size_t b = 0; b < h->n && rc == FPIC_OK; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        get_rand_kernel<T><<<blocks_for(h->n), 256, 0, h->stream>>>(arrays<T>(h, h->cur), h->n, b, m, srand, salive, scells,
                                                                  h->nr, h->nz);
        if (srand && h->spec.rng_mode == 1) // no stored state: what the next sub-step would draw for each particle
            counter_rand_readback_kernel<<<blocks_for(m), 256, 0, h->stream>>>(srand, b, m, h->spec.rng_seed_lo, h->spec.rng_seed_hi, h->t_substep);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && rand) e = hipMemcpyAsync(rand + 4 * b, srand, m * 4 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && alive) e = hipMemcpyAsync(alive + b, salive, m, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && cells) e = hipMemcpyAsync(cells + b, scells, m * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e));
    }
    if (srand) (void)hipFree(srand);
    if (salive) (void)hipFree(salive);
    if (scells) (void)hipFree(scells);
    return rc;
}

template <typename T>
int launch_precalc(fpic_handle* h)
{
    T h_, fr, fz, f_rz, f_zr;
    if (sizeof(T) == 4) {
        // u_h goes through uniform1f; the factors are decimal literals in the shader text
        h_ = static_cast<T>(h->k.h);
        fr = static_cast<T>(shader_literal(h->k.factor_r));
        fz = static_cast<T>(shader_literal(h->k.factor_z));
        f_rz = static_cast<T>(shader_literal(h->k.f_rz));
        f_zr = static_cast<T>(shader_literal(h->k.f_zr));
    } else {
        h_ = static_cast<T>(h->k.h); fr = static_cast<T>(h->k.factor_r); fz = static_cast<T>(h->k.factor_z);
        f_rz = static_cast<T>(h->k.f_rz); f_zr = static_cast<T>(h->k.f_zr);
    }
    timing_begin(h, KC_PRECALC);
    precalc_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(static_cast<const T*>(h->B), static_cast<const T*>(h->E),
                                                                 h->ncell, h_, fr, fz, f_rz, f_zr, h->spec.physical_a,
                                                                 static_cast<T*>(h->coef));
    timing_end(h);
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// queue the read-back of the scatter's spill counter into the next of the two lagged slots
int record_spill(fpic_handle* h)
{
    const int slot = static_cast<int>(h->deposit_seq++ & 1);
    HIP_TRY(h, hipMemcpyAsync(h->spilled_host + slot, h->spilled, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->spill_event[slot], h->stream));
    h->spill_pending[slot] = true;
    return FPIC_OK;
}

template <typename T>
int launch_push(fpic_handle* h, int nsub)
{
    PushArgs<T> a;
    a.slab = static_cast<T*>(h->slab[h->cur]);
    a.stride = h->n_pad;
    a.alive = h->alive[h->cur];
    a.coef = static_cast<const T*>(h->coef);
    a.sink_alive = h->sink_alive;
    a.inv_cdf_xy = static_cast<const T*>(h->inv_cdf_xy);
    a.entropy = static_cast<const T*>(h->entropy);
    a.nr = h->nr; a.nz = h->nz;
    a.step_factor = static_cast<T>(h->k.step_factor); // uniform1f(u_step_factor) (empic.js:852)
    a.n = h->n;
    a.nsub = nsub;
    a.id = h->id[h->cur];
    a.seed_lo = h->spec.rng_seed_lo; a.seed_hi = h->spec.rng_seed_hi;
    a.t0 = h->t_substep;
    a.raster_bits = h->spec.raster_subpixel_bits;
    const bool ctr = h->spec.rng_mode == 1;
    const size_t lanes = (h->n + Vec16<T>::N - 1) / Vec16<T>::N;
    // binned, fusion not switched off: the push also counts the particles per tile and, on a re-binning
    // launch, writes the sorted order itself; float state with unfused_deposit == 0 forms the per-cell
    // sums as well (for double the coefficient window alone fills the LDS, so the sums stay separate)
    const bool fuse = h->binned && h->spec.unfused_deposit != 1;
    const bool sums = fuse && sizeof(T) == 4 && h->spec.unfused_deposit == 0 && h->spec.shape == FPIC_SHAPE_REF11; // the fused sums are the sprite's
    const bool scatter = fuse && h->scatter_pending;
    TileArgs<T> t{};
    t.ntx = h->ntx; t.ntz = h->ntz; t.ntiles = h->ntiles;
    t.work = h->work2[h->wl]; t.nwork = h->nwork2[h->wl];
    t.cell_sums = static_cast<T*>(h->cell_sums); t.spilled = h->spilled; t.tile_count = h->tile_count;
    t.id = h->id[h->cur];
    t.dst_slab = static_cast<T*>(h->slab[h->cur ^ 1]); t.dst_alive = h->alive[h->cur ^ 1]; t.dst_id = h->id[h->cur ^ 1];
    t.dst_tile_start = h->tile_start2[h->wl ^ 1]; t.dst_tile_cursor = h->tile_cursor;
    t.chunk_census = h->chunk_census;
    t.census_valid = scatter && h->chunk_census_fresh ? 1 : 0; // (the launch before was an in-place fused push over this very work list)
    // TEST SWITCH (tests/test_gpu_parity.py: the count pass and the per-item census must reserve the same ranges): the
    // re-binning launch counts its chunk itself although the launch before left its census
    if (t.census_valid && std::getenv("FPIC_TEST_COUNT_PASS")) t.census_valid = 0;
    h->sums_fresh = h->census_fresh = false;
    h->chunk_census_fresh = false;
    h->scatter_pending = false;
    const unsigned grid = static_cast<unsigned>(h->work_cap);
    timing_begin(h, KC_PUSH);
    if (fuse) {
        HIP_TRY(h, hipMemsetAsync(h->tile_count, 0, sizeof(uint32_t) * h->ntiles, h->stream));
        if constexpr (sizeof(T) == 4) {
            const size_t gcells = sums_cells(h->nr, h->nz);
            if (sums) {
                HIP_TRY(h, hipMemsetAsync(h->cell_sums, 0, gcells * 4 * sizeof(T), h->stream)); // clear_color (empic.js:1476)
                HIP_TRY(h, hipMemsetAsync(h->spilled, 0, sizeof(unsigned long long), h->stream));
                constexpr size_t lds = push_tiles_lds_bytes<T, true>();
                if (scatter && ctr) push_tiles_kernel<T, true, true, true><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
                else if (scatter) push_tiles_kernel<T, true, true, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
                else if (ctr) push_tiles_kernel<T, true, false, true><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
                else push_tiles_kernel<T, true, false, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
            } else {
                constexpr size_t lds = push_tiles_lds_bytes<T, true, false>();
                if (scatter && ctr) push_tiles_kernel<T, true, true, true, false><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
                else if (scatter) push_tiles_kernel<T, true, true, false, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
                else if (ctr) push_tiles_kernel<T, true, false, true, false><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
                else push_tiles_kernel<T, true, false, false, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
            }
        } else {
            constexpr size_t lds = push_tiles_lds_bytes<T, true, false>();
            if (scatter && ctr) push_tiles_kernel<T, true, true, true, false><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
            else if (scatter) push_tiles_kernel<T, true, true, false, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
            else if (ctr) push_tiles_kernel<T, true, false, true, false><<<grid, push_threads<true>(), lds, h->stream>>>(a, t);
            else push_tiles_kernel<T, true, false, false, false><<<grid, push_threads<false>(), lds, h->stream>>>(a, t);
        }
    } else if (h->binned) { // the work list of the last binning is valid until the next one: the push is in place
        if (ctr) push_tiles_kernel<T, false, false, true><<<grid, push_threads<true>(), push_tiles_lds_bytes<T, false>(), h->stream>>>(a, t);
        else push_tiles_kernel<T, false, false, false><<<grid, push_threads<false>(), push_tiles_lds_bytes<T, false>(), h->stream>>>(a, t);
    } else {
        if (ctr) push_kernel<T, true><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
        else push_kernel<T, false><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
    }
    h->t_substep += static_cast<unsigned long long>(nsub);
    timing_end(h);
    HIP_TRY(h, hipGetLastError());
    if (fuse) {
        h->sums_fresh = sums;
        h->census_fresh = true;
        h->chunk_census_fresh = !scatter; // (a re-binning launch writes none: its items are those of the list it leaves)
        if (scatter) { // this launch was the binning: the other set and the other tables are live now
            h->cur ^= 1;
            h->wl ^= 1;
            h->deposits_since_bin = 0;
            h->last_spill = 0;
            h->spill_pending[0] = h->spill_pending[1] = false; // its own count was taken against the old windows
            h->sort_passes++;
        } else if (sums) {
            if (int rc = record_spill(h)) return rc;
        }
    }
    return FPIC_OK;
}

template <typename T>
int launch_bin(fpic_handle* h)
{
    const size_t shmem = static_cast<size_t>(h->ntiles) * sizeof(uint32_t);
    const unsigned nb = blocks_for(h->n, 256 * kBinPer);
    ParticleArrays<T> src = arrays<T>(h, h->cur), dst = arrays<T>(h, h->cur ^ 1);
    timing_begin(h, KC_SORT);
    HIP_TRY(h, hipMemsetAsync(h->tile_count, 0, sizeof(uint32_t) * h->ntiles, h->stream));
    bin_count_kernel<T><<<nb, 256, shmem, h->stream>>>(src, h->n, h->nr, h->nz, h->ntx, h->ntiles, h->tile_count);
    const int nw = h->wl ^ 1;
    bin_scan_kernel<<<1, 1024, 0, h->stream>>>(h->tile_count, h->ntiles, h->tile_start2[nw], h->tile_cursor, h->work2[nw], h->nwork2[nw]);
    // large populations: scatter staged through LDS, in two levels (by group of tiles, then by tile) when there are
    // many tiles; after two passes the data is back in the set it started in
    const bool staged = h->n >= h->two_level_min;
    bool two_level = false;
    if (staged) {
        uint32_t div = 1; // few tiles: one staged pass is enough (runs of >= 64 elements)
        while (h->ntiles > 64 && div * div < h->ntiles) ++div;
        two_level = div > 1;
        const uint32_t ncoarse = (h->ntiles + div - 1) / div;
        auto columns = [](const ParticleArrays<T>& a, const ParticleArrays<T>& b) {
            SortColumns<T, 10, true> c{};
            const T* s[10] = { a.x, a.y, a.z, a.vx, a.vy, a.vz, a.u1, a.u2, a.c1, a.c2 };
            T* d[10] = { b.x, b.y, b.z, b.vx, b.vy, b.vz, b.u1, b.u2, b.c1, b.c2 };
            for (int k = 0; k < 10; ++k) { c.src[k] = s[k]; c.dst[k] = d[k]; }
            c.src_byte = a.alive; c.dst_byte = b.alive; c.src_id = a.id; c.dst_id = b.id;
            return c;
        };
        const RzTileKey<T> key{ h->nr, h->nz, h->ntx, h->ntiles - 1 };
        const size_t lds = sort_scatter_lds(sizeof(T));
        auto kern = sort_scatter_kernel<T, 10, true, RzTileKey<T>>;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        sort_chunks_kernel<<<1, 1024, 0, h->stream>>>(h->tile_start2[nw], h->ntiles, div, ncoarse, h->coarse_cursor);
        kern<<<blocks_for(h->n, kSortChunk), kSortThreads, lds, h->stream>>>(columns(src, dst), h->n, key, h->ntiles, div, ncoarse, h->tile_start2[nw],
                                                                          h->tile_cursor, nullptr);
        if (two_level) {
            HIP_TRY(h, hipMemsetAsync(h->tile_cursor, 0, sizeof(uint32_t) * h->ntiles, h->stream));
            kern<<<blocks_for(h->n, kSortChunk) + ncoarse, kSortThreads, lds, h->stream>>>(columns(dst, src), h->n, key, h->ntiles, div, ncoarse,
                                                                                        h->tile_start2[nw], h->tile_cursor, h->coarse_cursor);
        }
    } else {
        bin_scatter_kernel<T><<<nb, 256, shmem, h->stream>>>(src, dst, h->n, h->nr, h->nz, h->ntx, h->ntiles, h->tile_start2[nw],
                                                           h->tile_cursor);
    }
    timing_end(h);
    HIP_TRY(h, hipGetLastError());
    if (!two_level) h->cur ^= 1;
    h->wl = nw;
    h->census_fresh = h->chunk_census_fresh = false; // tile_count now describes this binning, not a fused push
    h->scatter_pending = false;
    h->binned = true;
    h->deposits_since_bin = 0;
    h->last_spill = 0;
    h->spill_pending[0] = h->spill_pending[1] = false; // counts taken before this binning say nothing now
    h->sort_passes++;
    return FPIC_OK;
}

template <typename T>
int launch_cell_sums(fpic_handle* h)
{
    const size_t gcells = sums_cells(h->nr, h->nz);
    timing_begin(h, KC_DEPOSIT);
    HIP_TRY(h, hipMemsetAsync(h->cell_sums, 0, gcells * 4 * sizeof(T), h->stream)); // clear_color [0,0,0,0] (empic.js:1476)
    HIP_TRY(h, hipMemsetAsync(h->spilled, 0, sizeof(unsigned long long), h->stream));
    if (h->spec.shape == FPIC_SHAPE_CIC)
        cic_sums_kernel<T><<<static_cast<unsigned>(h->work_cap), kSumsThreads, kSumsLdsBytes, h->stream>>>(
            arrays<T>(h, h->cur), h->nr, h->nz, h->ntx, h->work2[h->wl], h->nwork2[h->wl], static_cast<T*>(h->cell_sums), h->spilled);
    else
        cell_sums_kernel<T><<<static_cast<unsigned>(h->work_cap), kSumsThreads, kSumsLdsBytes, h->stream>>>(
            arrays<T>(h, h->cur), h->nr, h->nz, h->ntx, h->work2[h->wl], h->nwork2[h->wl], static_cast<T*>(h->cell_sums), h->spilled,
            h->spec.raster_subpixel_bits);
    timing_end(h);
    HIP_TRY(h, hipGetLastError());
    return record_spill(h);
}

// Work queued on this handle's stream that touches moments/norm/avg must come after a finish stage
// that ran on a caller's stream.
int wait_external_finish(fpic_handle* h)
{
    if (h->finish_pending) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->finish_event, 0));
        h->finish_pending = false;
    }
    return FPIC_OK;
}

template <typename T>
int launch_stamp_finish(fpic_handle* h, const void* sums = nullptr, hipStream_t on = nullptr)
{
    dim3 grid((h->nr + 31) / 32, (h->nz + 31) / 32);
    const bool external = on != nullptr && on != h->stream;
    if (!external) {
        if (int rc = wait_external_finish(h)) return rc;
        timing_begin(h, KC_STAMP);
    } else {
        // the caller's stream follows whatever this handle has queued so far (an earlier finish stage,
        // a read-back of the density grids), not what it queues later
        HIP_TRY(h, hipEventRecord(h->order_event, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(on, h->order_event, 0));
        // ... and an earlier finish stage that is still pending on ANOTHER caller's stream: both
        // read-modify-write the running average
        if (h->finish_pending) HIP_TRY(h, hipStreamWaitEvent(on, h->finish_event, 0));
    }
    stamp_finish_kernel<T><<<grid, 256, 0, external ? on : h->stream>>>(static_cast<const T*>(sums ? sums : h->cell_sums), h->nr, h->nz,
                                                                      h->stamp, static_cast<T*>(h->moments), static_cast<T*>(h->norm),
                                                                      static_cast<T*>(h->avg), static_cast<T>(0.01), // u_ratio (empic.js:1083)
                                                                      h->spec.shape == FPIC_SHAPE_CIC ? 1 : 0);
    if (!external) timing_end(h);
    HIP_TRY(h, hipGetLastError());
    if (external) {
        HIP_TRY(h, hipEventRecord(h->finish_event, on));
        h->finish_pending = true;
    }
    return FPIC_OK;
}

template <typename T, typename In>
int set_grid_t(fpic_handle* h, int which, const In* host, int ncomp)
{
    In* stage = nullptr;
    const size_t bytes = h->ncell * ncomp * sizeof(In);
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), bytes));
    hipError_t e = hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        T* dst = static_cast<T*>(which == FPIC_GRID_E ? h->E : (which == FPIC_GRID_B ? h->B : h->sink));
        pack_grid_kernel<T, In><<<blocks_for(h->ncell), 256, 0, h->stream>>>(stage, h->nr, h->nz, ncomp, dst,
                                                                           which == FPIC_GRID_SINK_MASK ? h->sink_alive : nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "grid upload failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename T, typename Out>
int read_grid_t(fpic_handle* h, int which, Out* host)
{
    size_t cells = h->ncell;
    if (which == FPIC_READ_INV_CDF) cells = static_cast<size_t>(kCdfSide) * kCdfSide;
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), cells * 4 * sizeof(Out)));
    const unsigned nb = blocks_for(cells), nb4 = blocks_for(cells * 4);
    switch (which) {
    case FPIC_READ_MOMENTS: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->moments), stage, cells * 4); break;
    case FPIC_READ_NORM: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->norm), stage, cells * 4); break;
    case FPIC_READ_AVG: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->avg), stage, cells * 4); break;
    case FPIC_READ_B: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->B), stage, cells * 4); break;
    case FPIC_READ_E: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->E), stage, cells * 4); break;
    case FPIC_READ_SINK: convert_kernel<Out, T><<<nb4, 256, 0, h->stream>>>(static_cast<const T*>(h->sink), stage, cells * 4); break;
    case FPIC_READ_R1: case FPIC_READ_R2: case FPIC_READ_R3: case FPIC_READ_A:
        unpack_coef_kernel<T, Out><<<nb, 256, 0, h->stream>>>(static_cast<const T*>(h->coef), cells, which - FPIC_READ_R1, stage);
        break;
    case FPIC_READ_INV_CDF: unpack_xy_kernel<T, Out><<<nb, 256, 0, h->stream>>>(static_cast<const T*>(h->inv_cdf_xy), cells, stage); break;
    default: (void)hipFree(stage); return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown grid %d", which);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, stage, cells * 4 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "grid read-back failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename T>
int paint_uniform(fpic_handle* h, int kind, double value)
{
    add_uniform_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(static_cast<T*>(h->B), h->nr, h->nz, kind,
                                                                     static_cast<T>(value));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

template <typename T>
int paint_loop(fpic_handle* h, double r, double z, double current)
{
    if (!h->shapes_ready) { // the reference draws both shapes inside the factory (empic.js:333-345)
        loop_shape_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(static_cast<T>(0.5), h->nr, h->nz, static_cast<T*>(h->shape_half));
        loop_shape_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(static_cast<T>(0.1), h->nr, h->nz, static_cast<T*>(h->shape_tenth));
        HIP_TRY(h, hipGetLastError());
        h->shapes_ready = true;
    }
    current_loop_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(
        static_cast<T*>(h->B), static_cast<const T*>(h->shape_half), static_cast<const T*>(h->shape_tenth), h->nr, h->nz,
        static_cast<T>(r * h->k.factor_r), static_cast<T>(z * h->k.factor_z), static_cast<T>(current));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

template <typename T>
int create_state(fpic_handle* h)
{
    for (int s = 0; s < 2; ++s) {
        if (int rc = dev_alloc(h, &h->slab[s], 10 * h->n_pad * sizeof(T), &h->bytes_particles)) return rc;
        for (int a = 0; a < 10; ++a) h->part[s][a] = static_cast<T*>(h->slab[s]) + a * h->n_pad;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&h->alive[s]), h->n_pad, &h->bytes_particles)) return rc;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&h->id[s]), h->n_pad * sizeof(uint32_t), &h->bytes_particles)) return rc;
    }
    const size_t rgba = h->ncell * 4 * sizeof(T);
    const size_t gcells = sums_cells(h->nr, h->nz);
    struct { void** p; size_t bytes; } grids[] = {
        { &h->E, rgba }, { &h->B, rgba }, { &h->sink, rgba }, { &h->moments, rgba }, { &h->norm, rgba }, { &h->avg, rgba },
        { &h->shape_half, rgba }, { &h->shape_tenth, rgba },
        { &h->coef, h->ncell * 12 * sizeof(T) },
        { &h->cell_sums, gcells * 4 * sizeof(T) },
        { &h->inv_cdf_xy, static_cast<size_t>(kCdfSide) * kCdfSide * 2 * sizeof(T) },
        { &h->entropy, static_cast<size_t>(kEntropySide) * kEntropySide * 4 * sizeof(T) },
    };
    for (auto& g : grids)
        if (int rc = dev_alloc(h, g.p, g.bytes, &h->bytes_grid)) return rc;
    if (int rc = dev_alloc(h, reinterpret_cast<void**>(&h->sink_alive), h->ncell, &h->bytes_grid)) return rc;
    if (int rc = dev_alloc(h, reinterpret_cast<void**>(&h->stamp), kStampCells * sizeof(float), &h->bytes_grid)) return rc;

    float w[kStampCells];
    build_stamp(w);
    if (h->spec.shape == FPIC_SHAPE_CIC) { // the bilinear sums are the moments already: a one-cell stamp
        for (float& v : w) v = 0.0f;
        w[kStampCells / 2] = 1.0f;
    }
    HIP_TRY(h, hipMemcpyAsync(h->stamp, w, sizeof w, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));

    for (int s = 0; s < 2; ++s) {
        init_particles_kernel<T><<<blocks_for(h->n_pad), 256, 0, h->stream>>>(arrays<T>(h, s), h->n_pad);
        HIP_TRY(h, hipGetLastError());
    }
    // default entropy table and per-particle random state
    float* tmp = nullptr;
    const size_t ne = static_cast<size_t>(4) * kEntropySide * kEntropySide;
    const size_t chunk = 16u << 20;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&tmp), std::max(ne, std::min(chunk, h->n) * 4) * sizeof(float)));
    const unsigned long long seed = 0x5EEDF051ull ^ (reinterpret_cast<uintptr_t>(h) * 0x9E3779B97F4A7C15ull);
    default_random_kernel<<<blocks_for(ne), 256, 0, h->stream>>>(tmp, ne, seed);
    int rc = upload_entropy<T>(h, tmp);
    for (size_t b = 0; b < h->n && rc == FPIC_OK; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        default_random_kernel<<<blocks_for(m * 4), 256, 0, h->stream>>>(tmp, m * 4, seed + 0x1234567ull * (b + 1));
        set_rand_kernel<T><<<blocks_for(h->n), 256, 0, h->stream>>>(tmp, b, m, arrays<T>(h, h->cur), h->n);
        if (hipGetLastError() != hipSuccess) rc = fail(h, FPIC_ERR_HIP, "default random state failed");
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "state initialisation failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

void release(fpic_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    fcomm::release(h);
    fes::release(h);
    for (int s = 0; s < 2; ++s) {
        if (h->slab[s]) (void)hipFree(h->slab[s]);
        if (h->alive[s]) (void)hipFree(h->alive[s]);
        if (h->id[s]) (void)hipFree(h->id[s]);
    }
    void* bufs[] = { h->E, h->B, h->sink, h->sink_alive, h->inv_cdf_xy, h->entropy, h->coef, h->cell_sums, h->moments,
                     h->norm, h->avg, h->stamp, h->shape_half, h->shape_tenth, h->tile_count, h->tile_start2[0],
                     h->tile_start2[1], h->tile_cursor, h->coarse_cursor, h->nwork2[0], h->nwork2[1], h->work2[0], h->work2[1], h->spilled, h->chunk_census };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (h->spilled_host) (void)hipHostFree(h->spilled_host);
    for (hipEvent_t e : h->spill_event) if (e) (void)hipEventDestroy(e);
    if (h->finish_event) (void)hipEventDestroy(h->finish_event);
    if (h->order_event) (void)hipEventDestroy(h->order_event);
    for (PendingTiming& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}


// ---------------------------------------------------------------- checkpoint
// Flat binary file: header, then every particle array in the CALLER's order (x y z vx vy vz
// u1 u2 c1 c2 as T, alive as bytes), then the grid tables that cannot be rebuilt (E, B, sink,
// inverse CDF, entropy, running average).  The reference keeps its state only in GPU textures
// and never reads it back (utilities.js:701-711 exists but is unused): this is the read-back /
// resume path a host needs.
// 2 since the header's fpic_spec is the one of ABI 2: a file of an older library is refused by its version
constexpr uint32_t kRzCheckpointVersion = 2;
struct CheckpointHeader {
    char magic[8];          // "FPICCKP1"
    uint32_t version;       // kRzCheckpointVersion
    uint32_t precision;     // fpic_dtype of every T below
    uint64_t n;
    int32_t nr, nz;
    uint64_t t_substep;
    int32_t rng_mode;
    uint32_t reserved;
    fpic_spec spec;
};

template <typename T>
__global__ __launch_bounds__(256) void to_caller_order_kernel(const T* __restrict__ src, const uint32_t* __restrict__ id, size_t n,
                                                              size_t chunk_begin, size_t chunk_n, T* __restrict__ out)
{
    const size_t s = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t i = id[s];
    if (i >= chunk_begin && i < chunk_begin + chunk_n) out[i - chunk_begin] = src[s];
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* id, size_t n)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) id[i] = static_cast<uint32_t>(i);
}

template <typename T>
__global__ __launch_bounds__(256) void sink_mask_kernel(const T* __restrict__ sink_rgba, uint8_t* __restrict__ mask, size_t ncell)
{
    const size_t c = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < ncell) mask[c] = sink_rgba[4 * c] > static_cast<T>(0.5) ? 1 : 0;
}

struct FileCloser {
    FILE* f;
    ~FileCloser() { if (f) std::fclose(f); }
};

// one particle array (T or bytes), caller order, through a bounded staging buffer
template <typename E>
int save_array(fpic_handle* h, FILE* f, const E* dev_array, std::vector<unsigned char>& host)
{
    const size_t chunk = 16u << 20;
    E* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, h->n) * sizeof(E)));
    int rc = FPIC_OK;
    for (size_t b = 0; b < h->n && rc == FPIC_OK; b += chunk) {
        const size_t m = std::min(chunk, h->n - b);
        host.resize(m * sizeof(E));
        to_caller_order_kernel<E><<<blocks_for(h->n), 256, 0, h->stream>>>(dev_array, h->id[h->cur], h->n, b, m, stage);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host.data(), stage, m * sizeof(E), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, FPIC_ERR_HIP, "checkpoint read-back failed: %s", hipGetErrorString(e));
        else if (std::fwrite(host.data(), sizeof(E), m, f) != m) rc = fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    }
    (void)hipFree(stage);
    return rc;
}

int save_block(fpic_handle* h, FILE* f, const void* dev, size_t bytes, std::vector<unsigned char>& host)
{
    host.resize(bytes);
    HIP_TRY(h, hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (std::fwrite(host.data(), 1, bytes, f) != bytes) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    return FPIC_OK;
}

int load_block(fpic_handle* h, FILE* f, void* dev, size_t bytes, std::vector<unsigned char>& host)
{
    const size_t chunk = 256u << 20;
    for (size_t b = 0; b < bytes; b += chunk) {
        const size_t m = std::min(chunk, bytes - b);
        host.resize(m);
        if (std::fread(host.data(), 1, m, f) != m) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
        HIP_TRY(h, hipMemcpyAsync(static_cast<unsigned char*>(dev) + b, host.data(), m, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return FPIC_OK;
}

template <typename T>
int save_state(fpic_handle* h, FILE* f)
{
    std::vector<unsigned char> host;
    for (int a = 0; a < 10; ++a)
        if (int rc = save_array<T>(h, f, static_cast<const T*>(h->part[h->cur][a]), host)) return rc;
    if (int rc = save_array<uint8_t>(h, f, h->alive[h->cur], host)) return rc;
    const size_t rgba = h->ncell * 4 * sizeof(T);
    struct { const void* p; size_t bytes; } grids[] = {
        { h->E, rgba }, { h->B, rgba }, { h->sink, rgba },
        { h->inv_cdf_xy, static_cast<size_t>(kCdfSide) * kCdfSide * 2 * sizeof(T) },
        { h->entropy, static_cast<size_t>(kEntropySide) * kEntropySide * 4 * sizeof(T) },
        { h->avg, rgba },
    };
    for (auto& g : grids)
        if (int rc = save_block(h, f, g.p, g.bytes, host)) return rc;
    return FPIC_OK;
}

template <typename T>
int load_state(fpic_handle* h, FILE* f)
{
    std::vector<unsigned char> host;
    // the file is in the caller's order: it becomes the memory order, ids restart at identity
    for (int a = 0; a < 10; ++a)
        if (int rc = load_block(h, f, h->part[h->cur][a], h->n * sizeof(T), host)) return rc;
    if (int rc = load_block(h, f, h->alive[h->cur], h->n, host)) return rc;
    const size_t rgba = h->ncell * 4 * sizeof(T);
    struct { void* p; size_t bytes; } grids[] = {
        { h->E, rgba }, { h->B, rgba }, { h->sink, rgba },
        { h->inv_cdf_xy, static_cast<size_t>(kCdfSide) * kCdfSide * 2 * sizeof(T) },
        { h->entropy, static_cast<size_t>(kEntropySide) * kEntropySide * 4 * sizeof(T) },
        { h->avg, rgba },
    };
    for (auto& g : grids)
        if (int rc = load_block(h, f, g.p, g.bytes, host)) return rc;
    sink_mask_kernel<T><<<blocks_for(h->ncell), 256, 0, h->stream>>>(static_cast<const T*>(h->sink), h->sink_alive, h->ncell);
    HIP_TRY(h, hipGetLastError());
    return launch_precalc<T>(h); // the coefficient records follow from E and B
}

// out.set({source_pdf}) (empic.js:1263-1349): the 512x512 inverse-CDF table is built on the
// device (fpic_injection.hpp) into a scratch table and adopted only if the reference would
// not have thrown.
template <typename T, typename In>
int set_source_pdf(fpic_handle* h, const In* host_pdf)
{
    const size_t table_elems = static_cast<size_t>(2) * kCdfSide * kCdfSide;
    In* pdf = nullptr;
    double *row_cdf = nullptr, *row_sum = nullptr, *col_cdf = nullptr;
    T* table = nullptr;
    int* throws = nullptr;
    auto cleanup = [&]() {
        for (void* p : { static_cast<void*>(pdf), static_cast<void*>(row_cdf), static_cast<void*>(row_sum),
                         static_cast<void*>(col_cdf), static_cast<void*>(table), static_cast<void*>(throws) })
            if (p) (void)hipFree(p);
    };
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&pdf), h->ncell * sizeof(In));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&row_cdf), h->ncell * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&row_sum), h->nr * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&col_cdf), h->nr * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&table), table_elems * sizeof(T));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&throws), sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(throws, 0, sizeof(int), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(table, 0, table_elems * sizeof(T), h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(pdf, host_pdf, h->ncell * sizeof(In), hipMemcpyHostToDevice, h->stream);
    int flag = 0;
    if (e == hipSuccess) {
        cdf_rows_kernel<In><<<blocks_for(h->nr, 64), 64, 0, h->stream>>>(pdf, h->nr, h->nz, row_cdf, row_sum);
        cdf_cols_kernel<<<1, 1, 0, h->stream>>>(row_sum, h->nr, col_cdf);
        inverse_cdf_kernel<T><<<blocks_for(static_cast<size_t>(kCdfSide) * kCdfSide), 256, 0, h->stream>>>(col_cdf, row_cdf, h->nr, h->nz, table, throws);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, throws, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && !flag) {
        e = hipMemcpyAsync(h->inv_cdf_xy, table, table_elems * sizeof(T), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    cleanup();
    if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? FPIC_ERR_OOM : FPIC_ERR_HIP, "inverse-CDF build failed: %s", hipGetErrorString(e));
    if (flag) return fail(h, FPIC_ERR_INVALID_ARG, ".source_pdf <- the first grid row carries no weight (the reference throws a TypeError here)");
    return FPIC_OK;
}

template <typename K>
hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

// The reference validates with typeof === 'number' (utilities.js:118-127); over a C
// struct the equivalent failure is a non-finite or out-of-domain value.
int validate_spec(const fpic_spec* s)
{
    struct { const char* name; double v; bool positive; } fields[] = {
        { "radius", s->radius, true }, { "height", s->height, true }, { "dt", s->dt, false },
        { "particle_mass", s->particle_mass, true }, { "particle_charge", s->particle_charge, false },
    };
    for (auto& f : fields) {
        if (!std::isfinite(f.v)) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".%s <- Property does not match any given possible types!", f.name);
        if (f.positive && !(f.v > 0)) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".%s <- must be positive", f.name);
    }
    if (s->nr < 1) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".nr <- must be a positive integer");
    if (s->nz < 1) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".nz <- must be a positive integer");
    if (s->count == 0 && s->nparticles < 1) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".nparticles <- must be a positive integer");
    if (s->precision != FPIC_F32 && s->precision != FPIC_F64) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".precision <- must be 0 (f32) or 1 (f64)");
    if (s->rng_mode != 0 && s->rng_mode != 1) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".rng_mode <- must be 0 (reference) or 1 (counter)");
    if (s->unfused_deposit < 0 || s->unfused_deposit > 2) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".unfused_deposit <- must be 0, 1 or 2");
    if (s->shape != FPIC_SHAPE_REF11 && s->shape != FPIC_SHAPE_CIC) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".shape <- must be 0 (ref11) or 1 (cic)");
    if (s->raster_subpixel_bits < 0 || s->raster_subpixel_bits > 8) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".raster_subpixel_bits <- must be 0 (ideal sprites) or 1..8");
    if (s->raster_subpixel_bits && s->shape != FPIC_SHAPE_REF11) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".raster_subpixel_bits <- only the reference's point sprites are rasterised (shape 0)");
    if (s->geometry != FPIC_GEOM_CYL_RZ && s->geometry != FPIC_GEOM_CART3D) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".geometry <- must be 0 (cyl_rz) or 1 (cart3d)");
    return FPIC_OK;
}

} // namespace

// =============================================================================== ABI

// calls that exist only on the reference's (r,z) pusher / only on the CART3D box
#define RZ_ONLY(h, name)                                                                                            \
    do {                                                                                                            \
        if ((h)->es) return fail(h, FPIC_ERR_STATE, name " is not available on a CART3D handle (spec.geometry = 1)"); \
    } while (0)
#define BOX_ONLY(h, name)                                                                                           \
    do {                                                                                                            \
        if (!(h)->es) return fail(h, FPIC_ERR_STATE, name " needs a CART3D handle (spec.geometry = 1)");            \
    } while (0)

extern "C" {

const char* fpic_last_error(const fpic_handle* h) { return h ? h->err.c_str() : create_error().c_str(); }
int fpic_abi_version(void) { return FPIC_ABI_VERSION; }
const char* fpic_build_arch(void) { return "gfx950"; }

int fpic_create(const fpic_spec* spec, fpic_handle** out)
{
    if (!out) return fail(nullptr, FPIC_ERR_INVALID_ARG, "null output pointer");
    *out = nullptr;
    if (!spec) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".spec <- Non-optional property is undefined!");
    if (int rc = validate_spec(spec)) return rc;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, FPIC_ERR_NO_DEVICE, "no HIP device visible (%s): libfusionpic has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (spec->device < 0 || spec->device >= ndev) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".device <- ordinal %d out of range [0,%d)", spec->device, ndev);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, spec->device)) != hipSuccess)
        return fail(nullptr, FPIC_ERR_HIP, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FPIC_ERR_NO_DEVICE, "device %d is %s; libfusionpic carries gfx950 code objects only", spec->device, prop.gcnArchName);

    fpic_handle* h = new (std::nothrow) fpic_handle();
    if (!h) return fail(nullptr, FPIC_ERR_OOM, "host allocation failed");
    h->spec = *spec;
    if (const char* v = std::getenv("FPIC_TWO_LEVEL_MIN")) h->two_level_min = static_cast<size_t>(std::strtoull(v, nullptr, 10));
    h->k = derive_constants(*spec);
    h->prec = spec->precision;
    h->esize = spec->precision == FPIC_F64 ? 8 : 4;
    h->device = spec->device;
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    h->nr = spec->nr;
    h->nz = spec->nz;
    h->ncell = static_cast<size_t>(spec->nr) * spec->nz;
    h->n = spec->count ? static_cast<size_t>(spec->count) : static_cast<size_t>(spec->nparticles) * spec->nparticles; // empic.js:107-109
    h->n_pad = (h->n + 1023) / 1024 * 1024;
    h->ntx = (h->nr + 1 + kTileSide - 1) / kTileSide;
    h->ntz = (h->nz + 1 + kTileSide - 1) / kTileSide;
    h->ntiles = static_cast<uint32_t>(h->ntx) * h->ntz + 1;
    int rc = FPIC_OK;
    if (h->n >= 0xFFFFFFFFull - 4096) rc = fail(nullptr, FPIC_ERR_INVALID_ARG, ".nparticles <- at most 2^32 particles per device");
    else if (spec->geometry == FPIC_GEOM_CYL_RZ && h->ntiles > static_cast<uint32_t>(kMaxTiles)) rc = fail(nullptr, FPIC_ERR_INVALID_ARG, ".nr <- grid of %d x %d cells exceeds %d tiles of %d^2 cells", h->nr, h->nz, kMaxTiles, kTileSide);
    if (rc) { delete h; return rc; }

    auto bail = [&](int code) { create_error() = h->err; release(h); return code; };
    if (hipSetDevice(h->device) != hipSuccess) return bail(fail(h, FPIC_ERR_HIP, "hipSetDevice failed"));
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)));
    h->stream = h->own_stream;

    if (spec->geometry == FPIC_GEOM_CART3D) { // the electrostatic box: its own state and kernels (fes_api.hip)
        if ((rc = fes::create(h))) return bail(rc);
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(fail(h, FPIC_ERR_HIP, "handle setup failed: %s", hipGetErrorString(e)));
        *out = h;
        return FPIC_OK;
    }
    rc = (h->prec == FPIC_F32) ? create_state<float>(h) : create_state<double>(h);
    if (rc) return bail(rc);
    // the scatter's LDS image (72 KiB of double accumulators) exceeds the 64 KiB static limit
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(cic_sums_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSumsLdsBytes))) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(cic_sums_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSumsLdsBytes))) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(cell_sums_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSumsLdsBytes))) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void*>(cell_sums_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSumsLdsBytes))) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, false, false, false>, push_tiles_lds_bytes<float, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, false, false, true>, push_tiles_lds_bytes<float, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, false, false>, push_tiles_lds_bytes<float, true>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, false, true>, push_tiles_lds_bytes<float, true>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, true, false>, push_tiles_lds_bytes<float, true>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, true, true>, push_tiles_lds_bytes<float, true>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, false, false, false>, push_tiles_lds_bytes<float, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, false, true, false>, push_tiles_lds_bytes<float, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, true, false, false>, push_tiles_lds_bytes<float, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<float, true, true, true, false>, push_tiles_lds_bytes<float, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, true, false, false, false>, push_tiles_lds_bytes<double, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, true, false, true, false>, push_tiles_lds_bytes<double, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, true, true, false, false>, push_tiles_lds_bytes<double, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, true, true, true, false>, push_tiles_lds_bytes<double, true, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, false, false, false>, push_tiles_lds_bytes<double, false>())) != hipSuccess ||
        (e = set_lds(push_tiles_kernel<double, false, false, true>, push_tiles_lds_bytes<double, false>())) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e)));

    h->work_cap = (h->n + kDepositChunk - 1) / kDepositChunk + h->ntiles;
    uint64_t* acc = &h->bytes_grid;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&h->tile_count), sizeof(uint32_t) * h->ntiles, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->tile_start2[0]), sizeof(uint32_t) * (h->ntiles + 1), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->tile_start2[1]), sizeof(uint32_t) * (h->ntiles + 1), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->tile_cursor), sizeof(uint32_t) * h->ntiles, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->coarse_cursor), sizeof(uint32_t) * (kSortMaxBins + 1), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->nwork2[0]), sizeof(uint32_t), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->nwork2[1]), sizeof(uint32_t), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->work2[0]), sizeof(BlockWork) * h->work_cap, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->work2[1]), sizeof(BlockWork) * h->work_cap, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->chunk_census), sizeof(uint32_t) * kNbrSlots * h->work_cap, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&h->spilled), sizeof(unsigned long long), acc)))
        return bail(rc);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->spilled_host), 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->spill_event[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->spill_event[1], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->finish_event, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&h->order_event, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamSynchronize(h->stream)) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "handle setup failed: %s", hipGetErrorString(e)));
    h->spilled_host[0] = h->spilled_host[1] = 0;
    *out = h;
    return FPIC_OK;
}

int fpic_destroy(fpic_handle* h)
{
    if (!h) return FPIC_OK;
    release(h);
    return FPIC_OK;
}

int fpic_set_particles(fpic_handle* h, const void* pos_aos, const void* vel_aos, uint64_t n, int dtype)
{
    CHECK_HANDLE(h);
    if (h->es) {
        if (n != h->n) return fail(h, FPIC_ERR_INVALID_ARG, ".position <- expected %zu particles, got %llu", h->n, static_cast<unsigned long long>(n));
        return fes::set_particles(h, 0, pos_aos, vel_aos, 0, n, dtype);
    }
    if (n != h->n) return fail(h, FPIC_ERR_INVALID_ARG, ".position <- expected %zu particles, got %llu", h->n, static_cast<unsigned long long>(n));
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    const double fr = h->k.factor_r, fz = h->k.factor_z;
    int rc = FPIC_OK;
    for (int pass = 0; pass < 2 && rc == FPIC_OK; ++pass) {
        const void* src = pass == 0 ? pos_aos : vel_aos;
        if (!src) continue;
        const int first = pass == 0 ? 0 : 3;
        const bool al = pass == 0;
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? upload_vec3<float, float>(h, static_cast<const float*>(src), first, fr, fz, al)
                                   : upload_vec3<float, double>(h, static_cast<const double*>(src), first, fr, fz, al);
        else
            rc = dtype == FPIC_F32 ? upload_vec3<double, float>(h, static_cast<const float*>(src), first, fr, fz, al)
                                   : upload_vec3<double, double>(h, static_cast<const double*>(src), first, fr, fz, al);
    }
    if (pos_aos) { // positions changed under the bins
        h->binned = false;
        h->census_fresh = h->chunk_census_fresh = h->scatter_pending = false;
    }
    if (pos_aos || vel_aos) h->sums_fresh = false;
    return rc;
}

int fpic_set_grid(fpic_handle* h, int which, const void* data, int nr, int nz, int ncomp, int dtype)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_set_grid");
    if (!data) return fail(h, FPIC_ERR_INVALID_ARG, ".data <- Non-optional property is undefined!");
    if (nr != h->nr || nz != h->nz) return fail(h, FPIC_ERR_INVALID_ARG, ".grid <- expected %d x %d, got %d x %d", h->nr, h->nz, nr, nz);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    const bool vec = which == FPIC_GRID_E || which == FPIC_GRID_B;
    if (which < FPIC_GRID_E || which > FPIC_GRID_SOURCE_PDF) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown grid %d", which);
    if (ncomp != (vec ? 3 : 1)) return fail(h, FPIC_ERR_INVALID_ARG, ".ncomp <- expected %d", vec ? 3 : 1);

    if (which == FPIC_GRID_SOURCE_PDF)
        return h->prec == FPIC_F32 ? (dtype == FPIC_F32 ? set_source_pdf<float, float>(h, static_cast<const float*>(data))
                                                        : set_source_pdf<float, double>(h, static_cast<const double*>(data)))
                                   : (dtype == FPIC_F32 ? set_source_pdf<double, float>(h, static_cast<const float*>(data))
                                                        : set_source_pdf<double, double>(h, static_cast<const double*>(data)));
    if (h->prec == FPIC_F32)
        return dtype == FPIC_F32 ? set_grid_t<float, float>(h, which, static_cast<const float*>(data), ncomp)
                                 : set_grid_t<float, double>(h, which, static_cast<const double*>(data), ncomp);
    return dtype == FPIC_F32 ? set_grid_t<double, float>(h, which, static_cast<const float*>(data), ncomp)
                             : set_grid_t<double, double>(h, which, static_cast<const double*>(data), ncomp);
}

int fpic_set_random_state(fpic_handle* h, const float* entropy, const float* rand)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_set_random_state");
    if (h->spec.rng_mode == 1 && (entropy || rand))
        return fail(h, FPIC_ERR_STATE, ".rng_mode <- the counter-based generator has no entropy table or per-particle state; use rng_seed and fpic_set_substep_counter");
    if (entropy) {
        const size_t ne = static_cast<size_t>(4) * kEntropySide * kEntropySide;
        float* stage = nullptr;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), ne * sizeof(float)));
        hipError_t e = hipMemcpyAsync(stage, entropy, ne * sizeof(float), hipMemcpyHostToDevice, h->stream);
        int rc = FPIC_OK;
        if (e == hipSuccess) rc = h->prec == FPIC_F32 ? upload_entropy<float>(h, stage) : upload_entropy<double>(h, stage);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(stage);
        if (rc) return rc;
        if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "entropy upload failed: %s", hipGetErrorString(e));
    }
    if (rand) return h->prec == FPIC_F32 ? upload_rand_chunked<float>(h, rand, nullptr) : upload_rand_chunked<double>(h, rand, nullptr);
    return FPIC_OK;
}

int fpic_add_current_loop(fpic_handle* h, double r, double z, double current)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_add_current_loop");
    return h->prec == FPIC_F32 ? paint_loop<float>(h, r, z, current) : paint_loop<double>(h, r, z, current);
}
int fpic_add_current_z(fpic_handle* h, double current)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_add_current_z");
    return h->prec == FPIC_F32 ? paint_uniform<float>(h, 0, current) : paint_uniform<double>(h, 0, current);
}
int fpic_add_bz(fpic_handle* h, double bz)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::add_b(h, 0.0, 0.0, bz);
    return h->prec == FPIC_F32 ? paint_uniform<float>(h, 1, bz) : paint_uniform<double>(h, 1, bz);
}
int fpic_add_btheta(fpic_handle* h, double btheta)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_add_btheta");
    return h->prec == FPIC_F32 ? paint_uniform<float>(h, 2, btheta) : paint_uniform<double>(h, 2, btheta);
}

int fpic_precalc(fpic_handle* h)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::precalc(h);
    return h->prec == FPIC_F32 ? launch_precalc<float>(h) : launch_precalc<double>(h);
}

int fpic_substeps(fpic_handle* h, int nsub)
{
    CHECK_HANDLE(h);
    if (nsub < 0) return fail(h, FPIC_ERR_INVALID_ARG, ".ncalls <- must be >= 0");
    if (nsub == 0) return FPIC_OK;
    if (nsub > (1 << 30)) return fail(h, FPIC_ERR_INVALID_ARG, ".ncalls <- too large");
    if (h->es) return fes::substeps(h, nsub);
    const int rc = h->prec == FPIC_F32 ? launch_push<float>(h, nsub) : launch_push<double>(h, nsub);
    if (rc == FPIC_OK) {
        h->step_launches++;
        h->particle_updates += static_cast<uint64_t>(nsub) * h->n;
    }
    return rc;
}

int fpic_step(fpic_handle* h, int ncalls)
{
    if (ncalls > (1 << 29)) return fail(h, FPIC_ERR_INVALID_ARG, ".ncalls <- too large");
    return fpic_substeps(h, ncalls < 0 ? ncalls : 2 * ncalls);
}

int fpic_get_substep_counter(fpic_handle* h, uint64_t* t)
{
    CHECK_HANDLE(h);
    if (t) *t = h->t_substep;
    return FPIC_OK;
}

int fpic_set_substep_counter(fpic_handle* h, uint64_t t)
{
    CHECK_HANDLE(h);
    h->t_substep = t;
    return FPIC_OK;
}

int fpic_sort(fpic_handle* h)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::sort(h);
    return h->prec == FPIC_F32 ? launch_bin<float>(h) : launch_bin<double>(h);
}

int fpic_deposit(fpic_handle* h)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::density(h); // electrostatic: always at hand; full EM: deposited now
    bool rebin = !h->binned;
    if (!rebin) {
        if (h->spec.sort_interval > 0) {
            rebin = h->deposits_since_bin >= h->spec.sort_interval;
        } else {
            // adaptive: the slot about to be reused holds the count of the scatter two
            // frames back; waiting for it bounds the host's lead, it does not drain the GPU
            const int slot = static_cast<int>(h->deposit_seq & 1);
            if (h->spill_pending[slot]) {
                HIP_TRY(h, hipEventSynchronize(h->spill_event[slot]));
                h->last_spill = h->spilled_host[slot];
                h->spill_pending[slot] = false;
            }
            // the count roughly doubles per frame: 0.4 % two frames ago is about 2 % now
            rebin = h->last_spill * 256 > h->n;
        }
    }
    if (rebin) {
        if (h->binned && h->census_fresh && h->spec.unfused_deposit != 1) {
            // The last push counted the particles per tile as it stored them.  Lay the
            // next order out from that census now; the next push writes it (no extra pass).
            const int nw = h->wl ^ 1;
            timing_begin(h, KC_SORT);
            bin_scan_kernel<<<1, 1024, 0, h->stream>>>(h->tile_count, h->ntiles, h->tile_start2[nw], h->tile_cursor, h->work2[nw], h->nwork2[nw]);
            timing_end(h);
            HIP_TRY(h, hipGetLastError());
            h->scatter_pending = true;
            h->last_spill = 0; // decided; do not decide again on the same stale count
        } else if (int rc = fpic_sort(h)) {
            return rc;
        }
    }
    h->deposits_since_bin++;
    if (h->sums_fresh) return FPIC_OK; // the last step() already summed this very state (fused push)
    const int rc = h->prec == FPIC_F32 ? launch_cell_sums<float>(h) : launch_cell_sums<double>(h);
    if (rc == FPIC_OK) { h->deposit_launches++; h->sums_fresh = true; }
    return rc;
}

int fpic_density_finish(fpic_handle* h)
{
    CHECK_HANDLE(h);
    if (h->es) return FPIC_OK;
    return h->prec == FPIC_F32 ? launch_stamp_finish<float>(h) : launch_stamp_finish<double>(h);
}

int fpic_density_finish_from(fpic_handle* h, const void* sums, void* hip_stream)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_density_finish_from");
    if (!sums) return fail(h, FPIC_ERR_INVALID_ARG, ".sums <- Non-optional property is undefined!");
    hipStream_t on = static_cast<hipStream_t>(hip_stream);
    return h->prec == FPIC_F32 ? launch_stamp_finish<float>(h, sums, on) : launch_stamp_finish<double>(h, sums, on);
}

// density() of a rank of a sharded run: scatter, all-reduce (sum) of the per-cell sums over the ranks, finish
static int density_reduced(fpic_handle* h)
{
    fcomm::Comm* c = h->comm;
    const fdyn::Rccl& rc = fdyn::rccl();
    const size_t count = sums_cells(h->nr, h->nz) * 4;
    const ncclDataType_t dt = h->prec == FPIC_F32 ? ncclFloat : ncclDouble;
    if (!c->overlap) {
        if (int e = fcomm::check(h, rc.AllReduce(h->cell_sums, h->cell_sums, count, dt, ncclSum, c->nccl, h->stream), "ncclAllReduce")) return e;
        return fpic_density_finish(h);
    }
    if (!c->buf) {
        c->buf_bytes = count * h->esize;
        HIP_TRY(h, hipMalloc(&c->buf, c->buf_bytes));
    }
    // the previous frame's finish stage still reads buf: the copy waits for it, nothing else on the handle's stream does
    if (c->reduce_pending) HIP_TRY(h, hipStreamWaitEvent(h->stream, c->reduced, 0));
    HIP_TRY(h, hipMemcpyAsync(c->buf, h->cell_sums, c->buf_bytes, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(c->copied, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(c->side, c->copied, 0));
    if (int e = fcomm::check(h, rc.AllReduce(c->buf, c->buf, count, dt, ncclSum, c->nccl, c->side), "ncclAllReduce")) return e;
    const int rcode = h->prec == FPIC_F32 ? launch_stamp_finish<float>(h, c->buf, c->side) : launch_stamp_finish<double>(h, c->buf, c->side);
    if (rcode) return rcode;
    HIP_TRY(h, hipEventRecord(c->reduced, c->side));
    c->reduce_pending = true;
    return FPIC_OK;
}

int fpic_density(fpic_handle* h)
{
    if (int rc = fpic_deposit(h)) return rc;
    if (h->comm && !h->es) return density_reduced(h);
    return fpic_density_finish(h);
}

int fpic_read_grid(fpic_handle* h, int which, void* out, int dtype)
{
    CHECK_HANDLE(h);
    RZ_ONLY(h, "fpic_read_grid");
    if (int rc = wait_external_finish(h)) return rc;
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, ".out <- Non-optional property is undefined!");
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    if (h->prec == FPIC_F32)
        return dtype == FPIC_F32 ? read_grid_t<float, float>(h, which, static_cast<float*>(out))
                                 : read_grid_t<float, double>(h, which, static_cast<double*>(out));
    return dtype == FPIC_F32 ? read_grid_t<double, float>(h, which, static_cast<float*>(out))
                             : read_grid_t<double, double>(h, which, static_cast<double*>(out));
}

int fpic_get_particles(fpic_handle* h, void* pos_aos, void* vel_aos, float* rand, uint8_t* alive, int dtype)
{
    CHECK_HANDLE(h);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    if (h->es) { // the periodic box loses no particle and keeps no random state
        if (rand) std::memset(rand, 0, sizeof(float) * 4 * h->n);
        if (alive) std::memset(alive, 1, h->n);
        return fes::get_particles(h, 0, pos_aos, vel_aos, dtype);
    }
    int rc = FPIC_OK;
    for (int pass = 0; pass < 2 && rc == FPIC_OK; ++pass) {
        void* dst = pass == 0 ? pos_aos : vel_aos;
        if (!dst) continue;
        const int first = pass == 0 ? 0 : 3;
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? download_vec3<float, float>(h, static_cast<float*>(dst), first)
                                   : download_vec3<float, double>(h, static_cast<double*>(dst), first);
        else
            rc = dtype == FPIC_F32 ? download_vec3<double, float>(h, static_cast<float*>(dst), first)
                                   : download_vec3<double, double>(h, static_cast<double*>(dst), first);
    }
    if (rc == FPIC_OK && (rand || alive))
        rc = h->prec == FPIC_F32 ? download_misc<float>(h, rand, alive, nullptr) : download_misc<double>(h, rand, alive, nullptr);
    return rc;
}

int fpic_get_cells(fpic_handle* h, int32_t* cells)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::get_cells(h, 0, cells);
    if (!cells) return fail(h, FPIC_ERR_INVALID_ARG, ".cells <- Non-optional property is undefined!");
    return h->prec == FPIC_F32 ? download_misc<float>(h, nullptr, nullptr, cells) : download_misc<double>(h, nullptr, nullptr, cells);
}

int fpic_device_buffer(fpic_handle* h, int which, void** dptr, size_t* bytes)
{
    CHECK_HANDLE(h);
    if (h->es) return fes::device_buffer(h, which, dptr, bytes);
    if (which != FPIC_BUF_CELL_SUMS) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown buffer %d", which);
    if (dptr) *dptr = h->cell_sums;
    if (bytes) *bytes = sums_cells(h->nr, h->nz) * 4 * h->esize;
    return FPIC_OK;
}

int fpic_set_stream(fpic_handle* h, void* hip_stream)
{
    CHECK_HANDLE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    timing_collect(h);
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return FPIC_OK;
}

int fpic_get_stream(fpic_handle* h, void** hip_stream)
{
    CHECK_HANDLE(h);
    if (hip_stream) *hip_stream = h->stream;
    return FPIC_OK;
}

int fpic_save_checkpoint(fpic_handle* h, const char* path)
{
    CHECK_HANDLE(h);
    if (!path) return fail(h, FPIC_ERR_INVALID_ARG, ".path <- Non-optional property is undefined!");
    if (h->es) return fes::save_checkpoint(h, path);
    if (int rc = wait_external_finish(h)) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    FileCloser fc{ std::fopen(path, "wb") };
    if (!fc.f) return fail(h, FPIC_ERR_STATE, "cannot open %s for writing", path);
    CheckpointHeader hd{};
    std::memcpy(hd.magic, "FPICCKP1", 8);
    hd.version = kRzCheckpointVersion; hd.precision = static_cast<uint32_t>(h->prec); hd.n = h->n; hd.nr = h->nr; hd.nz = h->nz;
    hd.t_substep = h->t_substep; hd.rng_mode = h->spec.rng_mode; hd.spec = h->spec;
    if (std::fwrite(&hd, sizeof hd, 1, fc.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    return h->prec == FPIC_F32 ? save_state<float>(h, fc.f) : save_state<double>(h, fc.f);
}

int fpic_load_checkpoint(fpic_handle* h, const char* path)
{
    CHECK_HANDLE(h);
    if (!path) return fail(h, FPIC_ERR_INVALID_ARG, ".path <- Non-optional property is undefined!");
    if (h->es) return fes::load_checkpoint(h, path);
    if (int rc = wait_external_finish(h)) return rc;
    FileCloser fc{ std::fopen(path, "rb") };
    if (!fc.f) return fail(h, FPIC_ERR_STATE, "cannot open %s", path);
    CheckpointHeader hd{};
    // (magic and version are the first twelve bytes whatever the rest of the header looked like when the file was written)
    if (std::fread(&hd, 12, 1, fc.f) != 1 || std::memcmp(hd.magic, "FPICCKP1", 8) != 0)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is not a fusionpic checkpoint", path);
    if (hd.version != kRzCheckpointVersion && hd.version != 1) // (version 1: the same layout, written before the number was raised)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is a checkpoint of format version %u; this library reads version %u (the header embeds fpic_spec of ABI %d)", path, hd.version,
                    kRzCheckpointVersion, FPIC_ABI_VERSION);
    if (std::fread(reinterpret_cast<char*>(&hd) + 12, sizeof hd - 12, 1, fc.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
    if (hd.n != h->n || hd.nr != h->nr || hd.nz != h->nz || static_cast<int>(hd.precision) != h->prec || hd.rng_mode != h->spec.rng_mode)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint holds %llu particles on %d x %d, precision %u, rng %d; the pusher was made for %zu on %d x %d, precision %d, rng %d",
                    static_cast<unsigned long long>(hd.n), hd.nr, hd.nz, hd.precision, hd.rng_mode, h->n, h->nr, h->nz, h->prec, h->spec.rng_mode);
    if (hd.spec.radius != h->spec.radius || hd.spec.height != h->spec.height || hd.spec.dt != h->spec.dt ||
        hd.spec.particle_mass != h->spec.particle_mass || hd.spec.particle_charge != h->spec.particle_charge)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint was written with different radius/height/dt/mass/charge");
    // the whole payload must be there before any device state is touched: a short file is refused
    // with the handle unchanged
    const size_t rgba = h->ncell * 4 * h->esize;
    const unsigned long long want = sizeof hd + 10ull * h->n * h->esize + h->n + 4ull * rgba +
                                    static_cast<unsigned long long>(kCdfSide) * kCdfSide * 2 * h->esize +
                                    static_cast<unsigned long long>(kEntropySide) * kEntropySide * 4 * h->esize;
    if (std::fseek(fc.f, 0, SEEK_END) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    const long long have = std::ftell(fc.f);
    if (have < 0 || static_cast<unsigned long long>(have) < want)
        return fail(h, FPIC_ERR_STATE, "checkpoint is truncated: %lld bytes, %llu expected", have, want);
    if (std::fseek(fc.f, static_cast<long>(sizeof hd), SEEK_SET) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // from here on the particle arrays are overwritten in the caller's order: whatever the bins, the
    // census and the fused sums said about the old order is void, also if a read fails half-way
    h->binned = false;
    h->sums_fresh = h->census_fresh = h->chunk_census_fresh = h->scatter_pending = false;
    h->spill_pending[0] = h->spill_pending[1] = false;
    h->last_spill = 0;
    iota_kernel<<<blocks_for(h->n), 256, 0, h->stream>>>(h->id[h->cur], h->n);
    HIP_TRY(h, hipGetLastError());
    const int rc = h->prec == FPIC_F32 ? load_state<float>(h, fc.f) : load_state<double>(h, fc.f);
    if (rc) return rc;
    h->t_substep = hd.t_substep;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

// ---- CART3D extension entry points
int fpic_add_species(fpic_handle* h, double mass, double charge, uint64_t count, int* index)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_add_species");
    return fes::add_species(h, mass, charge, count, index);
}
int fpic_set_particles_of(fpic_handle* h, int species, const void* pos_aos, const void* vel_aos, uint64_t n, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_set_particles_of");
    const uint64_t have = fes::species_count(h, species);
    if (have != ~0ull && n != have) return fail(h, FPIC_ERR_INVALID_ARG, ".position <- expected %llu particles, got %llu", static_cast<unsigned long long>(have), static_cast<unsigned long long>(n));
    return fes::set_particles(h, species, pos_aos, vel_aos, 0, n, dtype);
}
int fpic_set_particles_range(fpic_handle* h, int species, uint64_t first, uint64_t n, const void* pos_aos, const void* vel_aos, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_set_particles_range");
    if (fes::is_decomposed(h)) return fail(h, FPIC_ERR_STATE, "a decomposed handle takes its particles through fpic_domain_set_particles");
    return fes::set_particles(h, species, pos_aos, vel_aos, first, n, dtype);
}

// ---- CART3D spatial decomposition (z-slabs)
int fpic_domain_init(fpic_handle* h, int rank, int world, int ghost_planes, int migrate_every, int distributed_solve)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_domain_init");
    return fes::domain_init(h, rank, world, ghost_planes, migrate_every, distributed_solve);
}
int fpic_domain_set_particles(fpic_handle* h, int species, uint64_t n, const void* pos_aos, const void* vel_aos, uint32_t first_id, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_domain_set_particles");
    return fes::domain_set_particles(h, species, n, pos_aos, vel_aos, first_id, dtype);
}
int fpic_domain_get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, uint32_t* ids, uint64_t capacity, uint64_t* n, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_domain_get_particles");
    return fes::domain_get_particles(h, species, pos_aos, vel_aos, ids, capacity, n, dtype);
}
int fpic_domain_stats(fpic_handle* h, uint64_t* migrated, uint64_t* lost)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_domain_stats");
    return fes::domain_stats(h, migrated, lost);
}
int fpic_group_precalc(fpic_handle** handles, int n)
{
    if (!handles || n < 1 || !handles[0]) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".handles <- Non-optional property is undefined!");
    CHECK_HANDLE(handles[0]);
    return fes::group_run(handles, n, 0, 0);
}
int fpic_group_density(fpic_handle** handles, int n)
{
    if (!handles || n < 1 || !handles[0]) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".handles <- Non-optional property is undefined!");
    CHECK_HANDLE(handles[0]);
    return fes::group_run(handles, n, 2, 0);
}
int fpic_group_step(fpic_handle** handles, int n, int ncalls)
{
    if (!handles || n < 1 || !handles[0]) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".handles <- Non-optional property is undefined!");
    CHECK_HANDLE(handles[0]);
    if (ncalls < 0 || ncalls > (1 << 29)) return fail(handles[0], FPIC_ERR_INVALID_ARG, ".ncalls <- out of range");
    return fes::group_run(handles, n, 1, ncalls);
}
int fpic_get_particles_of(fpic_handle* h, int species, void* pos_aos, void* vel_aos, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_get_particles_of");
    return fes::get_particles(h, species, pos_aos, vel_aos, dtype);
}
int fpic_get_cells_of(fpic_handle* h, int species, int32_t* cells)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_get_cells_of");
    return fes::get_cells(h, species, cells);
}
int fpic_get_particles_range(fpic_handle* h, int species, uint64_t first, uint64_t n, uint64_t stride, void* pos_aos, void* vel_aos, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_get_particles_range");
    return fes::get_particles(h, species, pos_aos, vel_aos, dtype, first, n, stride);
}
int fpic_get_cells_range(fpic_handle* h, int species, uint64_t first, uint64_t n, uint64_t stride, int32_t* cells)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_get_cells_range");
    return fes::get_cells(h, species, cells, first, n, stride);
}
int fpic_add_b(fpic_handle* h, double bx, double by, double bz)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_add_b");
    return fes::add_b(h, bx, by, bz);
}
int fpic_set_field3(fpic_handle* h, int which, const void* data, int nx, int ny, int nz, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_set_field3");
    return fes::set_field3(h, which, data, nx, ny, nz, dtype);
}
int fpic_read_field3(fpic_handle* h, int which, void* out, int dtype)
{
    CHECK_HANDLE(h);
    BOX_ONLY(h, "fpic_read_field3");
    return fes::read_field3(h, which, out, dtype);
}

int fpic_sync(fpic_handle* h)
{
    CHECK_HANDLE(h);
    if (int rc = wait_external_finish(h)) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int fpic_profile(fpic_handle* h, int enable)
{
    CHECK_HANDLE(h);
    timing_collect(h);
    h->profiling = enable != 0;
    return FPIC_OK;
}

int fpic_get_stats(fpic_handle* h, fpic_stats* out)
{
    CHECK_HANDLE(h);
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, ".out <- Non-optional property is undefined!");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    timing_collect(h);
    if (h->deposit_seq) { // stream is idle: the newest count is there
        const int newest = static_cast<int>((h->deposit_seq - 1) & 1);
        if (h->spill_pending[newest]) h->last_spill = h->spilled_host[newest];
    }
    std::memset(out, 0, sizeof *out);
    out->n_particles = h->es ? fes::particle_count(h) : h->n;
    out->particle_updates = h->particle_updates;
    out->step_launches = h->step_launches;
    out->deposit_launches = h->deposit_launches;
    out->sort_passes = h->sort_passes;
    out->deposit_spilled = h->es ? fes::last_spill(h) : h->last_spill;
    out->ms_push = h->ms[KC_PUSH];
    out->ms_deposit = h->ms[KC_DEPOSIT];
    out->ms_stamp = h->ms[KC_STAMP];
    out->ms_precalc = h->ms[KC_PRECALC];
    out->ms_sort = h->ms[KC_SORT];
    out->ms_solve = h->ms[KC_SOLVE];
    out->solve_launches = h->solve_launches;
    out->bytes_particle_state = h->bytes_particles;
    out->bytes_grid_state = h->bytes_grid;
    return FPIC_OK;
}

int fpic_reset_stats(fpic_handle* h)
{
    CHECK_HANDLE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    timing_collect(h);
    for (double& m : h->ms) m = 0;
    h->particle_updates = h->step_launches = h->deposit_launches = h->sort_passes = h->solve_launches = 0;
    return FPIC_OK;
}

} // extern "C"
