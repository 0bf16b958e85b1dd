// fes_api.hip — state and orchestration of the CART3D electrostatic extension
// (spec.geometry == FPIC_GEOM_CART3D): the self-consistent cycle push + deposit + solve of
// BASELINE.json configs[2..4] on one GPU.  The reference has no such mode (its fields are
// static, empic.js:1436-1505); where it has a counterpart the same contract is kept: dt is
// fixed at construction (empic.js:44), step() is two leap-frog sub-steps (empic.js:1436-1469),
// precalc() is the fields stage that must precede the first step (empic.js:1413-1434).
//
// One sub-step = one fused kernel per species (gather, Boris, drift, deposit of the new
// position) followed by the field solve: int64 -> T conversion, rocFFT real forward, the
// k-space kernel, rocFFT real inverse, the gradient kernel.  rocFFT is bound at run time
// (fpic_dyn.hpp); without it a handle with solver = POISSON_FFT cannot be created.
#include "fes_api.hpp"
#include "fes_kernels.hpp"
#include "fpic_dyn.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

using namespace fpic;

namespace fes {

struct Species {
    double mass = 0, charge = 0;
    int Z = 1;
    size_t n = 0, n_pad = 0;
    void* slab[2] = {};
    uint32_t* id[2] = {};
    int cur = 0;
    // two bin tables: [wl] describes the live particle order, [wl ^ 1] is laid out by the next binning
    // (which may be the next push, see rebin_pending)
    uint32_t *tile_count = nullptr, *tile_cursor = nullptr;
    uint32_t *tile_start2[2] = {}, *nwork2[2] = {};
    BlockWork* work2[2] = {};
    int wl = 0;
    size_t work_cap = 0;
    bool binned = false;
    bool census_fresh = false;  // tile_count holds the census of the current positions (written by the last push)
    bool rebin_pending = false; // tables [wl ^ 1] are laid out from that census: the next push re-bins
};

struct State {
    int nx = 0, ny = 0, nz = 0;
    double lx = 0, ly = 0, lz = 0, W = 1;
    size_t nodes = 0;
    int solver = FPIC_SOLVER_NONE;
    int ntx = 0, nty = 0, ntz = 0;
    uint32_t ntiles = 0; // + 1 always-empty bin (the scan kernel's clipped bin)
    long long* rho_fixed = nullptr;
    void *rho = nullptr, *hat = nullptr, *phi = nullptr, *E4 = nullptr;
    double* k2[3] = {};
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info_f = nullptr, info_i = nullptr;
    void *work_f = nullptr, *work_i = nullptr;
    double B0[3] = { 0, 0, 0 };
    unsigned long long* spilled = nullptr;
    unsigned long long* spilled_host = nullptr; // pinned, 2 lagged slots
    hipEvent_t spill_event[2] = {};
    bool spill_pending[2] = {};
    unsigned long long spill_seq = 0, last_spill = 0;
    int substeps_since_bin = 0;
    bool fields_ready = false;
    std::vector<Species> sp;
};

namespace {

constexpr double kSpeedOfLight = 2.998e8;   // empic.js:27
constexpr double kEps0 = 8.8541878128e-12;
constexpr double kPi = 3.14159265358979323846;

size_t total_particles(const State* st)
{
    size_t n = 0;
    for (const Species& s : st->sp) n += s.n;
    return n;
}

template <typename T>
int alloc_species(fpic_handle* h, Species& s)
{
    State* st = h->es;
    s.n_pad = (s.n + 1023) / 1024 * 1024;
    for (int k = 0; k < 2; ++k) {
        if (int rc = dev_alloc(h, &s.slab[k], 6 * s.n_pad * sizeof(T), &h->bytes_particles)) return rc;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&s.id[k]), s.n_pad * sizeof(uint32_t), &h->bytes_particles)) return rc;
        init3_kernel<T><<<blocks_for(s.n_pad), 256, 0, h->stream>>>(static_cast<T*>(s.slab[k]), s.n_pad, s.id[k]);
        HIP_TRY(h, hipGetLastError());
    }
    s.work_cap = (s.n + kChunk3 - 1) / kChunk3 + st->ntiles;
    uint64_t* acc = &h->bytes_grid;
    int rc;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&s.tile_count), sizeof(uint32_t) * st->ntiles, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&s.tile_cursor), sizeof(uint32_t) * st->ntiles, acc)))
        return rc;
    for (int k = 0; k < 2; ++k)
        if ((rc = dev_alloc(h, reinterpret_cast<void**>(&s.tile_start2[k]), sizeof(uint32_t) * (st->ntiles + 1), acc)) ||
            (rc = dev_alloc(h, reinterpret_cast<void**>(&s.nwork2[k]), sizeof(uint32_t), acc)) ||
            (rc = dev_alloc(h, reinterpret_cast<void**>(&s.work2[k]), sizeof(BlockWork) * s.work_cap, acc)))
            return rc;
    return FPIC_OK;
}

void free_species(Species& s)
{
    for (int k = 0; k < 2; ++k) {
        if (s.slab[k]) (void)hipFree(s.slab[k]);
        if (s.id[k]) (void)hipFree(s.id[k]);
    }
    for (void* p : { static_cast<void*>(s.tile_count), static_cast<void*>(s.tile_cursor), static_cast<void*>(s.tile_start2[0]),
                     static_cast<void*>(s.tile_start2[1]), static_cast<void*>(s.nwork2[0]), static_cast<void*>(s.nwork2[1]),
                     static_cast<void*>(s.work2[0]), static_cast<void*>(s.work2[1]) })
        if (p) (void)hipFree(p);
}

template <typename T>
Push3Args<T> push_args(fpic_handle* h, const Species& s)
{
    const State* st = h->es;
    Push3Args<T> a{};
    a.slab = static_cast<T*>(s.slab[s.cur]);
    a.stride = s.n_pad;
    a.n = s.n;
    a.E4 = static_cast<const T*>(st->E4);
    a.rho = reinterpret_cast<unsigned long long*>(st->rho_fixed);
    a.nx = st->nx; a.ny = st->ny; a.nz = st->nz;
    // derived in double, rounded once into T (es3d_oracle.py push_params)
    const double hh = s.charge * h->spec.dt / (2 * s.mass); // empic.js:44
    const double t[3] = { hh * st->B0[0], hh * st->B0[1], hh * st->B0[2] };
    const double t2 = (t[0] * t[0] + t[1] * t[1]) + t[2] * t[2];
    const double step = h->spec.dt * kSpeedOfLight;         // empic.js:852
    a.hc = static_cast<T>(hh / kSpeedOfLight);
    a.tx = static_cast<T>(t[0]); a.ty = static_cast<T>(t[1]); a.tz = static_cast<T>(t[2]);
    a.sx = static_cast<T>(2 * t[0] / (1 + t2)); a.sy = static_cast<T>(2 * t[1] / (1 + t2)); a.sz = static_cast<T>(2 * t[2] / (1 + t2));
    a.dx = static_cast<T>(step / st->lx); a.dy = static_cast<T>(step / st->ly); a.dz = static_cast<T>(step / st->lz);
    a.Z = s.Z;
    a.ntx = st->ntx; a.nty = st->nty; a.ntz = st->ntz;
    a.work = s.work2[s.wl]; a.nwork = s.nwork2[s.wl];
    a.spilled = st->spilled;
    a.tile_count = s.tile_count;
    a.id = s.id[s.cur];
    a.dst_slab = static_cast<T*>(s.slab[s.cur ^ 1]);
    a.dst_id = s.id[s.cur ^ 1];
    a.dst_tile_start = s.tile_start2[s.wl ^ 1];
    a.dst_tile_cursor = s.tile_cursor;
    return a;
}

template <typename T, bool DEPOSIT_ONLY>
int launch_push(fpic_handle* h, Species& s)
{
    State* st = h->es;
    const Push3Args<T> a = push_args<T>(h, s);
    const bool has_b = st->B0[0] != 0 || st->B0[1] != 0 || st->B0[2] != 0;
    if (s.n == 0) return FPIC_OK;
    if (s.binned) {
        const unsigned grid = static_cast<unsigned>(s.work_cap);
        constexpr size_t lds = push3_lds_bytes<T>();
        if constexpr (DEPOSIT_ONLY) {
            push3_tiles_kernel<T, false, true><<<grid, kPushThreads3, lds, h->stream>>>(a);
        } else {
            const bool rebin = s.rebin_pending;
            s.census_fresh = s.rebin_pending = false;
            HIP_TRY(h, hipMemsetAsync(s.tile_count, 0, sizeof(uint32_t) * st->ntiles, h->stream));
            if (rebin && has_b) push3_tiles_kernel<T, true, false, true><<<grid, kPushThreads3, lds, h->stream>>>(a);
            else if (rebin) push3_tiles_kernel<T, false, false, true><<<grid, kPushThreads3, lds, h->stream>>>(a);
            else if (has_b) push3_tiles_kernel<T, true, false><<<grid, kPushThreads3, lds, h->stream>>>(a);
            else push3_tiles_kernel<T, false, false><<<grid, kPushThreads3, lds, h->stream>>>(a);
            HIP_TRY(h, hipGetLastError());
            s.census_fresh = true;
            if (rebin) { // this launch was the binning: the other set and the other tables are live now
                s.cur ^= 1;
                s.wl ^= 1;
            }
        }
    } else {
        const size_t lanes = (s.n + Vec16<T>::N - 1) / Vec16<T>::N;
        if (has_b && !DEPOSIT_ONLY) push3_flat_kernel<T, true, DEPOSIT_ONLY><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
        else push3_flat_kernel<T, false, DEPOSIT_ONLY><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
        if (!DEPOSIT_ONLY) s.census_fresh = false;
    }
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// re-bin one species by tile, out of place (count, scan, scatter)
template <typename T>
int launch_bin(fpic_handle* h, Species& s)
{
    State* st = h->es;
    if (s.n == 0) { s.binned = true; return FPIC_OK; }
    const size_t shmem = static_cast<size_t>(st->ntiles) * sizeof(uint32_t);
    const unsigned nb = blocks_for(s.n, 256 * kBinPer3);
    const T* src = static_cast<const T*>(s.slab[s.cur]);
    T* dst = static_cast<T*>(s.slab[s.cur ^ 1]);
    const int nw = s.wl ^ 1;
    HIP_TRY(h, hipMemsetAsync(s.tile_count, 0, sizeof(uint32_t) * st->ntiles, h->stream));
    bin3_count_kernel<T><<<nb, 256, shmem, h->stream>>>(src, s.n_pad, s.n, st->nx, st->ny, st->nz, st->ntx, st->nty, st->ntiles, s.tile_count);
    bin_scan_kernel<<<1, 1024, 0, h->stream>>>(s.tile_count, st->ntiles, s.tile_start2[nw], s.tile_cursor, s.work2[nw], s.nwork2[nw], static_cast<uint32_t>(kChunk3));
    bin3_scatter_kernel<T><<<nb, 256, shmem, h->stream>>>(src, dst, s.n_pad, s.id[s.cur], s.id[s.cur ^ 1], s.n, st->nx, st->ny, st->nz, st->ntx,
                                                        st->nty, st->ntiles, s.tile_start2[nw], s.tile_cursor);
    HIP_TRY(h, hipGetLastError());
    s.cur ^= 1;
    s.wl = nw;
    s.binned = true;
    s.census_fresh = s.rebin_pending = false; // tile_count now describes this binning, not a push
    return FPIC_OK;
}

// Re-bin every species.  A species whose last push left a census of the current positions is not moved now:
// its next bin table is laid out from that census and the next push writes the sorted order itself (no
// extra pass); otherwise (first binning, positions uploaded since, `force`) the three-pass binning runs.
template <typename T>
int bin_all(fpic_handle* h, bool force)
{
    State* st = h->es;
    timing_begin(h, KC_SORT);
    int rc = FPIC_OK;
    for (Species& s : st->sp) {
        if (!force && s.binned && s.census_fresh && s.n) {
            const int nw = s.wl ^ 1;
            bin_scan_kernel<<<1, 1024, 0, h->stream>>>(s.tile_count, st->ntiles, s.tile_start2[nw], s.tile_cursor, s.work2[nw], s.nwork2[nw], static_cast<uint32_t>(kChunk3));
            if (hipGetLastError() != hipSuccess) { rc = fail(h, FPIC_ERR_HIP, "bin table scan failed"); break; }
            s.rebin_pending = true;
        } else if ((rc = launch_bin<T>(h, s))) {
            break;
        }
    }
    timing_end(h);
    if (rc) return rc;
    st->substeps_since_bin = 0;
    st->last_spill = 0;
    st->spill_pending[0] = st->spill_pending[1] = false;
    h->sort_passes++;
    return FPIC_OK;
}

int fft_status(fpic_handle* h, rocfft_status s, const char* what)
{
    if (s == rocfft_status_success) return FPIC_OK;
    return fail(h, FPIC_ERR_HIP, "%s failed (rocfft_status %d)", what, static_cast<int>(s));
}

// rho_fixed -> E4 (es3d_rho_real, es3d_poisson, es3d_gradient)
template <typename T>
int launch_solve(fpic_handle* h)
{
    State* st = h->es;
    timing_begin(h, KC_SOLVE);
    const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
    const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv); // q0 W / (2^42 dV)
    rho_real_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(st->rho_fixed, st->nodes, scale, static_cast<T*>(st->rho));
    HIP_TRY(h, hipGetLastError());
    if (st->solver == FPIC_SOLVER_POISSON_FFT) {
        const fdyn::RocFFT& ff = fdyn::rocfft();
        const int nxh = st->nx / 2 + 1;
        if (int rc = fft_status(h, ff.execution_info_set_stream(st->info_f, h->stream), "rocfft_execution_info_set_stream")) return rc;
        if (int rc = fft_status(h, ff.execution_info_set_stream(st->info_i, h->stream), "rocfft_execution_info_set_stream")) return rc;
        void* in_f[1] = { st->rho };
        void* out_f[1] = { st->hat };
        if (int rc = fft_status(h, ff.execute(st->fwd, in_f, out_f, st->info_f), "rocfft_execute (forward)")) return rc;
        const size_t modes = static_cast<size_t>(nxh) * st->ny * st->nz;
        kspace_kernel<T><<<blocks_for(modes), 256, 0, h->stream>>>(static_cast<T*>(st->hat), nxh, st->ny, st->nz, st->k2[0], st->k2[1], st->k2[2],
                                                                  1.0 / (kEps0 * static_cast<double>(st->nodes)));
        HIP_TRY(h, hipGetLastError());
        void* in_i[1] = { st->hat };
        void* out_i[1] = { st->phi };
        if (int rc = fft_status(h, ff.execute(st->inv, in_i, out_i, st->info_i), "rocfft_execute (inverse)")) return rc;
        gradient_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(
            static_cast<const T*>(st->phi), st->nx, st->ny, st->nz, static_cast<T>(1.0 / (2.0 * (st->lx / st->nx))),
            static_cast<T>(1.0 / (2.0 * (st->ly / st->ny))), static_cast<T>(1.0 / (2.0 * (st->lz / st->nz))), static_cast<T*>(st->E4));
        HIP_TRY(h, hipGetLastError());
    }
    timing_end(h);
    h->solve_launches++;
    return FPIC_OK;
}

template <typename T, bool DEPOSIT_ONLY>
int deposit_cycle(fpic_handle* h)
{
    State* st = h->es;
    timing_begin(h, DEPOSIT_ONLY ? KC_DEPOSIT : KC_PUSH);
    HIP_TRY(h, hipMemsetAsync(st->rho_fixed, 0, st->nodes * sizeof(long long), h->stream));
    HIP_TRY(h, hipMemsetAsync(st->spilled, 0, sizeof(unsigned long long), h->stream));
    int rc = FPIC_OK;
    for (Species& s : st->sp)
        if ((rc = launch_push<T, DEPOSIT_ONLY>(h, s))) break;
    timing_end(h);
    return rc;
}

template <typename T>
int substep(fpic_handle* h)
{
    State* st = h->es;
    // adaptive re-binning: the slot about to be reused holds the count of two sub-steps back
    bool rebin = false;
    for (const Species& s : st->sp) rebin |= !s.binned;
    if (!rebin) {
        if (h->spec.sort_interval > 0) {
            rebin = st->substeps_since_bin >= h->spec.sort_interval;
        } else {
            const int slot = static_cast<int>(st->spill_seq & 1);
            if (st->spill_pending[slot]) {
                HIP_TRY(h, hipEventSynchronize(st->spill_event[slot]));
                st->last_spill = st->spilled_host[slot];
                st->spill_pending[slot] = false;
            }
            // the count of deposits outside the LDS window grows slowly, then explodes once the bulk reaches the
            // halo (profiles/r02_c3_rebin_policy.txt: 0.002 %, 0.006 %, 0.08 %, 0.4 % after 4, 8, 10, 12 sub-steps
            // of the bench scene); a fused re-binning launch costs about a third more than an in-place one
            rebin = st->last_spill * 4096 > total_particles(st) || st->substeps_since_bin >= 8;
        }
    }
    if (rebin)
        if (int rc = bin_all<T>(h, false)) return rc;
    if (int rc = deposit_cycle<T, false>(h)) return rc;
    const int slot = static_cast<int>(st->spill_seq++ & 1);
    HIP_TRY(h, hipMemcpyAsync(st->spilled_host + slot, st->spilled, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(st->spill_event[slot], h->stream));
    st->spill_pending[slot] = true;
    st->substeps_since_bin++;
    h->step_launches++;
    h->particle_updates += total_particles(st);
    return launch_solve<T>(h);
}

// host holds the caller's particles [first, first + count)
template <typename T, typename In>
int upload_pos(fpic_handle* h, Species& s, const In* host, size_t first, size_t count)
{
    State* st = h->es;
    const size_t chunk = 8u << 20;
    In* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(In)));
    T* a = static_cast<T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            set_pos3_kernel<T, In><<<blocks_for(s.n), 256, 0, h->stream>>>(stage, first + b, m, 1 / st->lx, 1 / st->ly, 1 / st->lz, a, a + s.n_pad,
                                                                         a + 2 * s.n_pad, s.id[s.cur], s.n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_vel(fpic_handle* h, Species& s, const In* host, size_t first, size_t count)
{
    const size_t chunk = 8u << 20;
    In* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(In)));
    T* a = static_cast<T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            // velocities stay in units of c, unscaled
            set_vec3_kernel<T, In><<<blocks_for(s.n), 256, 0, h->stream>>>(stage, first + b, m, 1.0, 1.0, a + 3 * s.n_pad, a + 4 * s.n_pad,
                                                                         a + 5 * s.n_pad, nullptr, s.id[s.cur], s.n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename Out>
int download_vec3(fpic_handle* h, const Species& s, Out* host, int first)
{
    const size_t chunk = 8u << 20;
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, s.n) * 3 * sizeof(Out)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < s.n; b += chunk) {
        const size_t m = std::min(chunk, s.n - b);
        get_vec3_kernel<T, Out><<<blocks_for(s.n), 256, 0, h->stream>>>(a + first * s.n_pad, a + (first + 1) * s.n_pad, a + (first + 2) * s.n_pad,
                                                                      s.id[s.cur], s.n, b, m, stage);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host + 3 * b, stage, m * 3 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T>
int download_cells(fpic_handle* h, const Species& s, int32_t* cells)
{
    State* st = h->es;
    const size_t chunk = 16u << 20;
    int32_t* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, s.n) * sizeof(int32_t)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < s.n; b += chunk) {
        const size_t m = std::min(chunk, s.n - b);
        cells3_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(a, a + s.n_pad, a + 2 * s.n_pad, s.id[s.cur], s.n, b, m, st->nx, st->ny, st->nz, stage);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(cells + b, stage, m * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "cell read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_field(fpic_handle* h, const In* host)
{
    State* st = h->es;
    In* stage = nullptr;
    const size_t bytes = st->nodes * 3 * sizeof(In);
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), bytes));
    hipError_t e = hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        pack_field3_kernel<T, In><<<blocks_for(st->nodes), 256, 0, h->stream>>>(stage, st->nx, st->ny, st->nz, static_cast<T*>(st->E4));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "field upload failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename T, typename Out>
int download_grid(fpic_handle* h, const void* dev, size_t count, Out* host)
{
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), count * sizeof(Out)));
    convert_kernel<Out, T><<<blocks_for(count), 256, 0, h->stream>>>(static_cast<const T*>(dev), stage, count);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, stage, count * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "grid read-back failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename K>
hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

template <typename T>
int create_state(fpic_handle* h)
{
    State* st = h->es;
    uint64_t* acc = &h->bytes_grid;
    const int nxh = st->nx / 2 + 1;
    int rc;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&st->rho_fixed), st->nodes * sizeof(long long), acc)) ||
        (rc = dev_alloc(h, &st->rho, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->phi, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->E4, st->nodes * 4 * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->hat, static_cast<size_t>(nxh) * st->ny * st->nz * 2 * sizeof(T), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&st->spilled), sizeof(unsigned long long), acc)))
        return rc;
    const int dims[3] = { st->nx, st->ny, st->nz };
    const double len[3] = { st->lx, st->ly, st->lz };
    for (int a = 0; a < 3; ++a) {
        std::vector<double> t(dims[a]);
        const double d = len[a] / dims[a];
        for (int l = 0; l < dims[a]; ++l) {
            const double s = 2.0 / d * std::sin(kPi * l / dims[a]); // es3d_k2_table
            t[l] = s * s;
        }
        if ((rc = dev_alloc(h, reinterpret_cast<void**>(&st->k2[a]), sizeof(double) * dims[a], acc))) return rc;
        HIP_TRY(h, hipMemcpyAsync(st->k2[a], t.data(), sizeof(double) * dims[a], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    hipError_t e;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&st->spilled_host), 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&st->spill_event[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&st->spill_event[1], hipEventDisableTiming)) != hipSuccess)
        return fail(h, FPIC_ERR_HIP, "handle setup failed: %s", hipGetErrorString(e));
    st->spilled_host[0] = st->spilled_host[1] = 0;
    if ((e = set_lds(push3_tiles_kernel<T, false, false>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, true, false>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, false, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, true, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(bin3_count_kernel<T>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_scatter_kernel<T>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess)
        return fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));

    if (st->solver == FPIC_SOLVER_POISSON_FFT) {
        const fdyn::RocFFT& ff = fdyn::rocfft();
        if (!ff.ok) return fail(h, FPIC_ERR_STATE, ".solver <- rocFFT is not available (%s); there is no other Poisson solver and no CPU fallback", ff.why.c_str());
        const size_t lengths[3] = { static_cast<size_t>(st->nx), static_cast<size_t>(st->ny), static_cast<size_t>(st->nz) };
        const rocfft_precision prec = sizeof(T) == 4 ? rocfft_precision_single : rocfft_precision_double;
        if ((rc = fft_status(h, ff.plan_create(&st->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 3, lengths, 1, nullptr), "rocfft_plan_create (forward)")) ||
            (rc = fft_status(h, ff.plan_create(&st->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 3, lengths, 1, nullptr), "rocfft_plan_create (inverse)")) ||
            (rc = fft_status(h, ff.execution_info_create(&st->info_f), "rocfft_execution_info_create")) ||
            (rc = fft_status(h, ff.execution_info_create(&st->info_i), "rocfft_execution_info_create")))
            return rc;
        size_t wf = 0, wi = 0;
        if ((rc = fft_status(h, ff.plan_get_work_buffer_size(st->fwd, &wf), "rocfft_plan_get_work_buffer_size")) ||
            (rc = fft_status(h, ff.plan_get_work_buffer_size(st->inv, &wi), "rocfft_plan_get_work_buffer_size")))
            return rc;
        if (wf) {
            if ((rc = dev_alloc(h, &st->work_f, wf, acc))) return rc;
            if ((rc = fft_status(h, ff.execution_info_set_work_buffer(st->info_f, st->work_f, wf), "rocfft_execution_info_set_work_buffer"))) return rc;
        }
        if (wi) {
            if ((rc = dev_alloc(h, &st->work_i, wi, acc))) return rc;
            if ((rc = fft_status(h, ff.execution_info_set_work_buffer(st->info_i, st->work_i, wi), "rocfft_execution_info_set_work_buffer"))) return rc;
        }
    }
    if ((rc = alloc_species<T>(h, st->sp[0]))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int check_species(fpic_handle* h, int species)
{
    if (species < 0 || species >= static_cast<int>(h->es->sp.size()))
        return fail(h, FPIC_ERR_INVALID_ARG, ".species <- %d is not one of the handle's %zu species", species, h->es->sp.size());
    return FPIC_OK;
}

} // namespace

uint64_t particle_count(const fpic_handle* h) { return total_particles(h->es); }
uint64_t last_spill(const fpic_handle* h) { return h->es->last_spill; }
uint64_t species_count(const fpic_handle* h, int species)
{
    return species >= 0 && species < static_cast<int>(h->es->sp.size()) ? h->es->sp[species].n : ~0ull;
}

int create(fpic_handle* h)
{
    const fpic_spec& sp = h->spec;
    if (sp.ny < 1) return fail(h, FPIC_ERR_INVALID_ARG, ".ny <- must be a positive integer");
    if (!(sp.length_y > 0) || !std::isfinite(sp.length_y)) return fail(h, FPIC_ERR_INVALID_ARG, ".length_y <- must be positive");
    if (sp.solver != FPIC_SOLVER_NONE && sp.solver != FPIC_SOLVER_POISSON_FFT) return fail(h, FPIC_ERR_INVALID_ARG, ".solver <- must be 0 (none) or 1 (poisson_fft)");
    if (sp.macro_weight < 0 || !std::isfinite(sp.macro_weight)) return fail(h, FPIC_ERR_INVALID_ARG, ".macro_weight <- must be positive");
    if (sp.particle_charge == 0) return fail(h, FPIC_ERR_INVALID_ARG, ".particle_charge <- must not be zero: it is the unit of the charge grid");
    if (sp.rng_mode != 0) return fail(h, FPIC_ERR_INVALID_ARG, ".rng <- the periodic box has no re-injection and no generator");
    State* st = new (std::nothrow) State();
    if (!st) return fail(h, FPIC_ERR_OOM, "host allocation failed");
    h->es = st;
    st->nx = sp.nr; st->ny = sp.ny; st->nz = sp.nz;
    st->lx = sp.radius; st->ly = sp.length_y; st->lz = sp.height;
    st->W = sp.macro_weight > 0 ? sp.macro_weight : 1.0;
    st->solver = sp.solver;
    st->fields_ready = sp.solver == FPIC_SOLVER_NONE; // static fields (zero until fpic_set_field3) need no precalc()
    st->nodes = static_cast<size_t>(st->nx) * st->ny * st->nz;
    if (st->nodes >= (1ull << 31)) return fail(h, FPIC_ERR_INVALID_ARG, ".nr <- at most 2^31 nodes per device");
    st->ntx = (st->nx + kTX - 1) / kTX;
    st->nty = (st->ny + kTY - 1) / kTY;
    st->ntz = (st->nz + kTZ - 1) / kTZ;
    const size_t nt = static_cast<size_t>(st->ntx) * st->nty * st->ntz + 1;
    if (nt > static_cast<size_t>(kMaxTiles3))
        return fail(h, FPIC_ERR_INVALID_ARG, ".nr <- grid of %d x %d x %d nodes exceeds %d tiles of %dx%dx%d cells per device", st->nx, st->ny, st->nz,
                    kMaxTiles3, kTX, kTY, kTZ);
    st->ntiles = static_cast<uint32_t>(nt);
    Species s0;
    s0.mass = sp.particle_mass; s0.charge = sp.particle_charge; s0.Z = 1;
    s0.n = h->n;
    st->sp.push_back(s0);
    return h->prec == FPIC_F32 ? create_state<float>(h) : create_state<double>(h);
}

void release(fpic_handle* h)
{
    State* st = h->es;
    if (!st) return;
    for (Species& s : st->sp) free_species(s);
    for (void* p : { static_cast<void*>(st->rho_fixed), st->rho, st->hat, st->phi, st->E4, static_cast<void*>(st->k2[0]), static_cast<void*>(st->k2[1]),
                     static_cast<void*>(st->k2[2]), st->work_f, st->work_i, static_cast<void*>(st->spilled) })
        if (p) (void)hipFree(p);
    if (st->spilled_host) (void)hipHostFree(st->spilled_host);
    for (hipEvent_t e : st->spill_event) if (e) (void)hipEventDestroy(e);
    const fdyn::RocFFT& ff = fdyn::rocfft();
    if (ff.ok) {
        if (st->fwd) (void)ff.plan_destroy(st->fwd);
        if (st->inv) (void)ff.plan_destroy(st->inv);
        if (st->info_f) (void)ff.execution_info_destroy(st->info_f);
        if (st->info_i) (void)ff.execution_info_destroy(st->info_i);
    }
    delete st;
    h->es = nullptr;
}

int add_species(fpic_handle* h, double mass, double charge, uint64_t count, int* index)
{
    State* st = h->es;
    if (!(mass > 0) || !std::isfinite(mass)) return fail(h, FPIC_ERR_INVALID_ARG, ".mass <- must be positive");
    const double z = charge / h->spec.particle_charge;
    const double zr = std::nearbyint(z);
    if (!std::isfinite(z) || std::fabs(z - zr) > 1e-6 || zr == 0 || std::fabs(zr) > 255)
        return fail(h, FPIC_ERR_INVALID_ARG, ".charge <- must be a non-zero integer multiple (|Z| <= 255) of spec.particle_charge");
    if (count >= 0xFFFFFFFFull - 4096) return fail(h, FPIC_ERR_INVALID_ARG, ".count <- at most 2^32 particles per species and device");
    Species s;
    s.mass = mass; s.charge = charge; s.Z = static_cast<int>(zr); s.n = static_cast<size_t>(count);
    st->sp.push_back(s);
    const int rc = h->prec == FPIC_F32 ? alloc_species<float>(h, st->sp.back()) : alloc_species<double>(h, st->sp.back());
    if (rc) { free_species(st->sp.back()); st->sp.pop_back(); return rc; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (index) *index = static_cast<int>(st->sp.size()) - 1;
    return FPIC_OK;
}

int set_particles(fpic_handle* h, int species, const void* pos_aos, const void* vel_aos, uint64_t first, uint64_t n, int dtype)
{
    if (int rc = check_species(h, species)) return rc;
    Species& s = h->es->sp[species];
    if (first > s.n || n > s.n - first)
        return fail(h, FPIC_ERR_INVALID_ARG, ".position <- particles [%llu, %llu) do not lie within the species' %zu", static_cast<unsigned long long>(first),
                    static_cast<unsigned long long>(first + n), s.n);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    int rc = FPIC_OK;
    if (pos_aos && n) {
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? upload_pos<float, float>(h, s, static_cast<const float*>(pos_aos), first, n) : upload_pos<float, double>(h, s, static_cast<const double*>(pos_aos), first, n);
        else
            rc = dtype == FPIC_F32 ? upload_pos<double, float>(h, s, static_cast<const float*>(pos_aos), first, n) : upload_pos<double, double>(h, s, static_cast<const double*>(pos_aos), first, n);
        s.binned = false;
        s.census_fresh = s.rebin_pending = false;
        if (h->es->solver != FPIC_SOLVER_NONE) h->es->fields_ready = false; // the fields of these positions are not known yet
    }
    if (rc == FPIC_OK && vel_aos && n) {
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? upload_vel<float, float>(h, s, static_cast<const float*>(vel_aos), first, n) : upload_vel<float, double>(h, s, static_cast<const double*>(vel_aos), first, n);
        else
            rc = dtype == FPIC_F32 ? upload_vel<double, float>(h, s, static_cast<const float*>(vel_aos), first, n) : upload_vel<double, double>(h, s, static_cast<const double*>(vel_aos), first, n);
    }
    return rc;
}

int get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, int dtype)
{
    if (int rc = check_species(h, species)) return rc;
    const Species& s = h->es->sp[species];
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    int rc = FPIC_OK;
    for (int pass = 0; pass < 2 && rc == FPIC_OK; ++pass) {
        void* dst = pass == 0 ? pos_aos : vel_aos;
        if (!dst || !s.n) continue;
        const int first = pass == 0 ? 0 : 3;
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? download_vec3<float, float>(h, s, static_cast<float*>(dst), first) : download_vec3<float, double>(h, s, static_cast<double*>(dst), first);
        else
            rc = dtype == FPIC_F32 ? download_vec3<double, float>(h, s, static_cast<float*>(dst), first) : download_vec3<double, double>(h, s, static_cast<double*>(dst), first);
    }
    return rc;
}

int get_cells(fpic_handle* h, int species, int32_t* cells)
{
    if (int rc = check_species(h, species)) return rc;
    if (!cells) return fail(h, FPIC_ERR_INVALID_ARG, ".cells <- Non-optional property is undefined!");
    const Species& s = h->es->sp[species];
    if (!s.n) return FPIC_OK;
    return h->prec == FPIC_F32 ? download_cells<float>(h, s, cells) : download_cells<double>(h, s, cells);
}

int add_b(fpic_handle* h, double bx, double by, double bz)
{
    if (!std::isfinite(bx) || !std::isfinite(by) || !std::isfinite(bz)) return fail(h, FPIC_ERR_INVALID_ARG, ".B <- must be finite");
    h->es->B0[0] += bx; h->es->B0[1] += by; h->es->B0[2] += bz;
    return FPIC_OK;
}

int set_field3(fpic_handle* h, int which, const void* data, int nx, int ny, int nz, int dtype)
{
    State* st = h->es;
    if (!data) return fail(h, FPIC_ERR_INVALID_ARG, ".data <- Non-optional property is undefined!");
    if (which != FPIC_F3_E) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- only E (0) can be uploaded");
    if (nx != st->nx || ny != st->ny || nz != st->nz) return fail(h, FPIC_ERR_INVALID_ARG, ".grid <- expected %d x %d x %d, got %d x %d x %d", st->nx, st->ny, st->nz, nx, ny, nz);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    st->fields_ready = true;
    if (h->prec == FPIC_F32)
        return dtype == FPIC_F32 ? upload_field<float, float>(h, static_cast<const float*>(data)) : upload_field<float, double>(h, static_cast<const double*>(data));
    return dtype == FPIC_F32 ? upload_field<double, float>(h, static_cast<const float*>(data)) : upload_field<double, double>(h, static_cast<const double*>(data));
}

int read_field3(fpic_handle* h, int which, void* out, int dtype)
{
    State* st = h->es;
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, ".out <- Non-optional property is undefined!");
    if (which == FPIC_F3_RHO_FIXED) {
        HIP_TRY(h, hipMemcpyAsync(out, st->rho_fixed, st->nodes * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return FPIC_OK;
    }
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    const void* dev;
    size_t count = st->nodes;
    switch (which) {
    case FPIC_F3_E: dev = st->E4; count *= 4; break;
    case FPIC_F3_RHO: dev = st->rho; break;
    case FPIC_F3_PHI: dev = st->phi; break;
    default: return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown grid %d", which);
    }
    if (h->prec == FPIC_F32)
        return dtype == FPIC_F32 ? download_grid<float, float>(h, dev, count, static_cast<float*>(out)) : download_grid<float, double>(h, dev, count, static_cast<double*>(out));
    return dtype == FPIC_F32 ? download_grid<double, float>(h, dev, count, static_cast<float*>(out)) : download_grid<double, double>(h, dev, count, static_cast<double*>(out));
}

int precalc(fpic_handle* h)
{
    int rc = h->prec == FPIC_F32 ? deposit_cycle<float, true>(h) : deposit_cycle<double, true>(h);
    if (rc) return rc;
    h->deposit_launches++;
    rc = h->prec == FPIC_F32 ? launch_solve<float>(h) : launch_solve<double>(h);
    if (rc == FPIC_OK) h->es->fields_ready = true;
    return rc;
}

int step(fpic_handle* h, int ncalls)
{
    if (!h->es->fields_ready)
        return fail(h, FPIC_ERR_STATE, "step() before precalc(): the fields of the current particle positions have not been computed");
    for (int k = 0; k < 2 * ncalls; ++k)
        if (int rc = h->prec == FPIC_F32 ? substep<float>(h) : substep<double>(h)) return rc;
    return FPIC_OK;
}

int sort(fpic_handle* h) { return h->prec == FPIC_F32 ? bin_all<float>(h, true) : bin_all<double>(h, true); }

int device_buffer(fpic_handle* h, int which, void** dptr, size_t* bytes)
{
    if (which != FPIC_BUF_RHO_FIXED) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown buffer %d", which);
    if (dptr) *dptr = h->es->rho_fixed;
    if (bytes) *bytes = h->es->nodes * sizeof(long long);
    return FPIC_OK;
}

} // namespace fes
