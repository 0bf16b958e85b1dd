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
//
// One translation unit, in pieces (round 5; no symbol changed): this file holds the entry points fpic_api.hip calls (argument
// and state checks, dispatch on precision) and the in-process group; the pieces it includes, in order —
//   fes_state.inc.hpp       Species, State, Domain
//   fes_host_push.inc.hpp   allocation of a species, the tiled push's launches, the binning passes
//   fes_host_solve.inc.hpp  the Poisson solve of one handle, the electrostatic sub-step
//   fes_host_em.inc.hpp     the full-EM cycle of one handle
//   fes_host_io.inc.hpp     uploads, downloads, creation of the state
//   fes_checkpoint.inc.hpp  checkpoint files (box, rank of a decomposition)
//   fes_domain.inc.hpp      the z-slab decomposition: message lists, both transports, migration, decomposed solves and cycles
#include "fes_api.hpp"
#include "fes_kernels.hpp"
#include "fes_fft.hpp"
#include "fes_tri.hpp"
#include "fpic_comm.hpp"
#include "fpic_dyn.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace fpic;

namespace fes {
#include "fes_state.inc.hpp"

namespace {
#include "fes_host_push.inc.hpp"
#include "fes_host_solve.inc.hpp"
#include "fes_host_em.inc.hpp"
#include "fes_host_io.inc.hpp"

} // namespace

uint64_t particle_count(const fpic_handle* h) { return total_particles(h->es); }
uint64_t last_spill(const fpic_handle* h) { return h->es->last_spill; }
bool is_decomposed(const fpic_handle* h) { return h->es && h->es->dom != nullptr; }
uint64_t species_count(const fpic_handle* h, int species)
{
    return species >= 0 && species < static_cast<int>(h->es->sp.size()) ? h->es->sp[species].n : ~0ull;
}

int create(fpic_handle* h)
{
    const fpic_spec& sp = h->spec;
    if (sp.ny < 1) return fail(h, FPIC_ERR_INVALID_ARG, ".ny <- must be a positive integer");
    if (!(sp.length_y > 0) || !std::isfinite(sp.length_y)) return fail(h, FPIC_ERR_INVALID_ARG, ".length_y <- must be positive");
    if (sp.solver != FPIC_SOLVER_NONE && sp.solver != FPIC_SOLVER_POISSON_FFT && sp.solver != FPIC_SOLVER_YEE)
        return fail(h, FPIC_ERR_INVALID_ARG, ".solver <- must be 0 (none), 1 (poisson_fft) or 2 (yee)");
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
    st->zs0 = 0; st->nzs = st->nz; // (a rank of a decomposition may keep its slab and halo only: domain_init)
    if (st->nodes >= (1ull << 31)) return fail(h, FPIC_ERR_INVALID_ARG, ".nr <- at most 2^31 nodes per device");
    if (sp.solver == FPIC_SOLVER_YEE) { // (the full-EM tile: 8 x 8 x 8 cells in float, 8 x 4 x 8 in double — EmWin, fes_kernels.hpp)
        st->ltx = st->ltz = kEL;
        st->lty = h->prec == FPIC_F32 ? EmWin<float>::LY : EmWin<double>::LY;
    }
    st->ntx = (st->nx + (1 << st->ltx) - 1) >> st->ltx;
    st->nty = (st->ny + (1 << st->lty) - 1) >> st->lty;
    st->ntz = (st->nz + (1 << st->ltz) - 1) >> st->ltz;
    const size_t nt = static_cast<size_t>(st->ntx) * st->nty * st->ntz + 1;
    if (nt > static_cast<size_t>(kMaxTilesStaged3))
        return fail(h, FPIC_ERR_INVALID_ARG, ".nr <- grid of %d x %d x %d nodes exceeds %d tiles of %dx%dx%d cells per device", st->nx, st->ny, st->nz,
                    kMaxTilesStaged3, 1 << st->ltx, 1 << st->lty, 1 << st->ltz);
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
    if (Domain* d = st->dom) {
        for (void* p : { static_cast<void*>(d->ghost_recv[0]), static_cast<void*>(d->ghost_recv[1]), d->mig_send[0], d->mig_send[1], d->mig_recv[0],
                         d->mig_recv[1], static_cast<void*>(d->counts_dev), static_cast<void*>(d->j_recv[0]), static_cast<void*>(d->j_recv[1]) })
            if (p) (void)hipFree(p);
        if (d->counts_host) (void)hipHostFree(d->counts_host);
        if (d->comm_stream) { (void)hipStreamSynchronize(d->comm_stream); (void)hipStreamDestroy(d->comm_stream); }
        if (d->ev_boundary) (void)hipEventDestroy(d->ev_boundary);
        if (d->ev_ghost) (void)hipEventDestroy(d->ev_ghost);
        for (void* p : { d->hatA, d->hatB, d->xbuf, d->hatZ, d->tri, d->fft_work[0], d->fft_work[1], d->fft_work[2], d->fft_work[3] })
            if (p) (void)hipFree(p);
        const fdyn::RocFFT& ffd = fdyn::rocfft();
        if (ffd.ok) {
            for (rocfft_plan pl : { d->p2f, d->p2i, d->pzf, d->pzi }) if (pl) (void)ffd.plan_destroy(pl);
            for (rocfft_execution_info in : { d->i2f, d->i2i, d->izf, d->izi }) if (in) (void)ffd.execution_info_destroy(in);
        }
        delete d;
    }
    for (void* p : { st->Ey, st->By, st->B4n, st->Bh[0], st->Bh[1], static_cast<void*>(st->Jfix), st->fft_tw[0], st->fft_tw[1], st->fft_tw[2] })
        if (p) (void)hipFree(p);
    for (void* p : { static_cast<void*>(st->rho_fixed), st->rho, st->hat, st->phi, st->E4, static_cast<void*>(st->k2[0]), static_cast<void*>(st->k2[1]),
                     static_cast<void*>(st->k2[2]), st->work_f, st->work_i, static_cast<void*>(st->spilled), static_cast<void*>(st->joint_work),
                     static_cast<void*>(st->joint_nwork) })
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
    if (st->dom)
        return fail(h, FPIC_ERR_STATE, "fpic_add_species after fpic_domain_init: the decomposition sized its message buffers from the species it knew; add every species first");
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
        s.census_fresh = s.rebin_pending = s.chunk_census_fresh = false;
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

// the caller's particles first, first + stride, ..., n of them (n = ~0: all of them from `first` on at that stride)
static int check_range(fpic_handle* h, const Species& s, uint64_t first, uint64_t& n, uint64_t stride)
{
    if (stride < 1) return fail(h, FPIC_ERR_INVALID_ARG, ".stride <- must be at least 1");
    const uint64_t most = first < s.n ? (s.n - first + stride - 1) / stride : 0;
    if (n == ~0ull) n = most;
    if (n > most) return fail(h, FPIC_ERR_INVALID_ARG, ".n <- %llu particles from %llu at stride %llu: the species has %llu", static_cast<unsigned long long>(n),
                              static_cast<unsigned long long>(first), static_cast<unsigned long long>(stride), static_cast<unsigned long long>(s.n));
    return FPIC_OK;
}

int get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, int dtype, uint64_t from, uint64_t n, uint64_t stride)
{
    if (h->es->dom) return fail(h, FPIC_ERR_STATE, "a decomposed handle holds a changing subset of the particles: read it with fpic_domain_get_particles");
    if (int rc = check_species(h, species)) return rc;
    const Species& s = h->es->sp[species];
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    if (int rc = check_range(h, s, from, n, stride)) return rc;
    int rc = FPIC_OK;
    for (int pass = 0; pass < 2 && rc == FPIC_OK; ++pass) {
        void* dst = pass == 0 ? pos_aos : vel_aos;
        if (!dst || !s.n || !n) continue;
        const int first = pass == 0 ? 0 : 3;
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? download_vec3<float, float>(h, s, static_cast<float*>(dst), first, from, n, stride) : download_vec3<float, double>(h, s, static_cast<double*>(dst), first, from, n, stride);
        else
            rc = dtype == FPIC_F32 ? download_vec3<double, float>(h, s, static_cast<float*>(dst), first, from, n, stride) : download_vec3<double, double>(h, s, static_cast<double*>(dst), first, from, n, stride);
    }
    return rc;
}

int get_cells(fpic_handle* h, int species, int32_t* cells, uint64_t from, uint64_t n, uint64_t stride)
{
    if (h->es->dom) return fail(h, FPIC_ERR_STATE, "a decomposed handle holds a changing subset of the particles: read it with fpic_domain_get_particles");
    if (int rc = check_species(h, species)) return rc;
    if (!cells) return fail(h, FPIC_ERR_INVALID_ARG, ".cells <- Non-optional property is undefined!");
    const Species& s = h->es->sp[species];
    if (int rc = check_range(h, s, from, n, stride)) return rc;
    if (!s.n || !n) return FPIC_OK;
    return h->prec == FPIC_F32 ? download_cells<float>(h, s, cells, from, n, stride) : download_cells<double>(h, s, cells, from, n, stride);
}

#include "fes_checkpoint.inc.hpp"

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
    const bool lattice = which == FPIC_F3_EDGE_E || which == FPIC_F3_FACE_B;
    if (which != FPIC_F3_E && !lattice) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- only E (0), EDGE_E (5) and FACE_B (6) can be uploaded");
    if (lattice && st->solver != FPIC_SOLVER_YEE) return fail(h, FPIC_ERR_STATE, ".which <- the lattice fields exist in the full-EM mode only (spec.solver = 2)");
    if (nx != st->nx || ny != st->ny || nz != st->nz) return fail(h, FPIC_ERR_INVALID_ARG, ".grid <- expected %d x %d x %d, got %d x %d x %d", st->nx, st->ny, st->nz, nx, ny, nz);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    st->fields_ready = true;
    if (int rc0 = em_close_any(h)) return rc0;  // (an upload of one lattice field leaves the other at the integer time)
    void* target = which == FPIC_F3_E ? st->E4 : (which == FPIC_F3_EDGE_E ? st->Ey : st->By);
    int rc;
    if (h->prec == FPIC_F32)
        rc = dtype == FPIC_F32 ? upload_field<float, float>(h, static_cast<const float*>(data), target) : upload_field<float, double>(h, static_cast<const double*>(data), target);
    else
        rc = dtype == FPIC_F32 ? upload_field<double, float>(h, static_cast<const float*>(data), target) : upload_field<double, double>(h, static_cast<const double*>(data), target);
    if (rc == FPIC_OK && lattice) rc = h->prec == FPIC_F32 ? em_nodes<float>(h) : em_nodes<double>(h);
    return rc;
}

// host side of a read-back from a rank with slab-only arrays: plane l of the held array is plane zs0 + l of the grid
template <typename V>
void spread_held(const State* st, const V* held, size_t per, V* out)
{
    std::memset(out, 0, per * st->nz * sizeof(V));
    for (int l = 0; l < st->nzs; ++l) {
        const int g = ((st->zs0 + l) % st->nz + st->nz) % st->nz;
        std::memcpy(out + g * per, held + static_cast<size_t>(l) * per, per * sizeof(V));
    }
}

int read_field3(fpic_handle* h, int which, void* out, int dtype)
{
    State* st = h->es;
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, ".out <- Non-optional property is undefined!");
    if (which == FPIC_F3_FACE_B)
        if (int rc = em_close_any(h)) return rc;   // (B of the integer time is formed when somebody asks for it)
    if (which == FPIC_F3_RHO_FIXED || which == FPIC_F3_J_FIXED) {
        if (which == FPIC_F3_J_FIXED && !st->Jfix) return fail(h, FPIC_ERR_STATE, ".which <- the current grid exists in the full-EM mode only (spec.solver = 2)");
        const void* src = which == FPIC_F3_RHO_FIXED ? static_cast<const void*>(st->rho_fixed) : static_cast<const void*>(st->Jfix);
        const size_t per = static_cast<size_t>(st->nx) * st->ny * (which == FPIC_F3_RHO_FIXED ? 1 : 3);
        if (compact(st)) {
            std::vector<long long> held(per * st->nzs);
            HIP_TRY(h, hipMemcpyAsync(held.data(), src, held.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            spread_held(st, held.data(), per, static_cast<long long*>(out));
            return FPIC_OK;
        }
        HIP_TRY(h, hipMemcpyAsync(out, src, per * st->nz * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return FPIC_OK;
    }
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    const void* dev;
    size_t count = st->nodes;
    switch (which) {
    case FPIC_F3_E: dev = st->E4; count *= 4; break;
    case FPIC_F3_RHO:
        if (int rc = h->prec == FPIC_F32 ? refresh_rho<float>(h) : refresh_rho<double>(h)) return rc;
        dev = st->rho; break;
    case FPIC_F3_PHI: dev = st->phi; break;
    case FPIC_F3_B_NODES: dev = st->B4n; count *= 4; break;
    case FPIC_F3_EDGE_E: dev = st->Ey; count *= 4; break;
    case FPIC_F3_FACE_B: dev = st->By; count *= 4; break;
    default: return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown grid %d", which);
    }
    if (!dev) return fail(h, FPIC_ERR_STATE, ".which <- this grid exists in the full-EM mode only (spec.solver = 2)");
    if (compact(st)) { // the planes this rank holds, in their places of a whole grid whose other planes read zero
        const size_t per = count / st->nz, held = per * st->nzs;
        int rc;
        if (dtype == FPIC_F32) {
            std::vector<float> tmp(held);
            rc = h->prec == FPIC_F32 ? download_grid<float, float>(h, dev, held, tmp.data()) : download_grid<double, float>(h, dev, held, tmp.data());
            if (rc == FPIC_OK) spread_held(st, tmp.data(), per, static_cast<float*>(out));
        } else {
            std::vector<double> tmp(held);
            rc = h->prec == FPIC_F32 ? download_grid<float, double>(h, dev, held, tmp.data()) : download_grid<double, double>(h, dev, held, tmp.data());
            if (rc == FPIC_OK) spread_held(st, tmp.data(), per, static_cast<double*>(out));
        }
        return rc;
    }
    if (h->prec == FPIC_F32)
        return dtype == FPIC_F32 ? download_grid<float, float>(h, dev, count, static_cast<float*>(out)) : download_grid<float, double>(h, dev, count, static_cast<double*>(out));
    return dtype == FPIC_F32 ? download_grid<double, float>(h, dev, count, static_cast<float*>(out)) : download_grid<double, double>(h, dev, count, static_cast<double*>(out));
}

// ================================================================ spatial decomposition (z-slabs)

namespace {
#include "fes_domain.inc.hpp"

} // namespace

int precalc(fpic_handle* h)
{
    if (h->es->dom) {
        Ranks rk;
        if (int e = dom_ranks_of(h, rk)) return e;
        return h->prec == FPIC_F32 ? dom_precalc<float>(rk) : dom_precalc<double>(rk);
    }
    if (h->es->solver == FPIC_SOLVER_YEE) {
        const int rc = h->prec == FPIC_F32 ? em_precalc<float>(h) : em_precalc<double>(h);
        if (rc == FPIC_OK) h->es->fields_ready = true;
        return rc;
    }
    // A freshly uploaded population is in the caller's order: deposited as it is, every particle adds its eight weights with
    // global atomics (369 ms at 2e9 particles, profiles/r03_bench_kernel_stats.csv: a third of that bench's GPU time).  Large
    // populations are binned first — the staged two-level scatter of 4.7, which the first sub-step would run anyway — and take
    // the tiled deposit; the charge grid is an integer grid, so the result is the same whatever the order.  Small ones keep
    // the flat form (nothing to gain, and the tests keep covering it).
    bool bin_first = false;
    for (const Species& sp : h->es->sp) bin_first |= !sp.binned && sp.n >= h->two_level_min;
    if (bin_first)
        if (int e = h->prec == FPIC_F32 ? bin_all<float>(h, false) : bin_all<double>(h, false)) return e;
    int rc = h->prec == FPIC_F32 ? deposit_cycle<float, true>(h) : deposit_cycle<double, true>(h);
    if (rc) return rc;
    h->deposit_launches++;
    rc = h->prec == FPIC_F32 ? launch_solve<float>(h) : launch_solve<double>(h);
    if (rc == FPIC_OK) h->es->fields_ready = true;
    return rc;
}

// density() of the box: the charge grid of the CURRENT positions (the EM cycle deposits currents, not charge)
int density(fpic_handle* h)
{
    if (h->es->solver != FPIC_SOLVER_YEE) return FPIC_OK; // the electrostatic cycle deposits the charge every sub-step
    if (h->es->dom) { // every rank of the decomposition calls it: the ghost planes of the charge grid are exchanged
        Ranks rk;
        if (int e = dom_ranks_of(h, rk)) return e;
        return h->prec == FPIC_F32 ? dom_density<float>(rk) : dom_density<double>(rk);
    }
    const int rc = h->prec == FPIC_F32 ? deposit_cycle<float, true>(h) : deposit_cycle<double, true>(h);
    if (rc == FPIC_OK) { h->deposit_launches++; h->es->rho_fresh = false; }
    return rc;
}

int substeps(fpic_handle* h, int nsub)
{
    if (!h->es->fields_ready)
        return fail(h, FPIC_ERR_STATE, "step() before precalc(): the fields of the current particle positions have not been computed");
    if (h->es->dom) {
        Ranks rk;
        if (int e = dom_ranks_of(h, rk)) return e;
        for (int k = 0; k < nsub; ++k)
            if (int rc = h->prec == FPIC_F32 ? dom_substep<float>(rk) : dom_substep<double>(rk)) return rc;
        return FPIC_OK;
    }
    if (h->es->solver == FPIC_SOLVER_YEE) {
        for (int k = 0; k < nsub; ++k)
            if (int rc = h->prec == FPIC_F32 ? em_substep<float>(h) : em_substep<double>(h)) return rc;
        return FPIC_OK;
    }
    for (int k = 0; k < nsub; ++k)
        if (int rc = h->prec == FPIC_F32 ? substep<float>(h) : substep<double>(h)) return rc;
    return FPIC_OK;
}

int sort(fpic_handle* h) { return h->prec == FPIC_F32 ? bin_all<float>(h, true) : bin_all<double>(h, true); }

int device_buffer(fpic_handle* h, int which, void** dptr, size_t* bytes)
{
    if (which != FPIC_BUF_RHO_FIXED) return fail(h, FPIC_ERR_INVALID_ARG, ".which <- unknown buffer %d", which);
    if (dptr) *dptr = h->es->rho_fixed;
    if (bytes) *bytes = held_nodes(h->es) * sizeof(long long); // (a rank with slab-only arrays: its nzs planes from zs0 on)
    return FPIC_OK;
}

// ---- decomposition entry points (fpic_domain_*, fpic_group_*)

// A rank whose cycle never touches a node outside its slab and `halo` planes on either side (+ 1 above: nodes, not
// cells) gives back its whole-grid node arrays and keeps nzs = nzl + 2 halo + 1 planes of each, from plane z0 - halo on
// (State::zs0, nzs; kernels map a plane through held_plane(), the host through lp()).  At 512^3 on eight ranks that is
// 77 planes of 512: 0.15 of the grid (and the whole-grid transform buffer goes: the decomposed solve has its own).
// FPIC_DOMAIN_COMPACT=0 keeps the whole-grid arrays (a development switch: both layouts run the same kernels).
template <typename T>
int keep_slab_only(fpic_handle* h, int halo)
{
    State* st = h->es;
    const Domain& d = *st->dom;
    const int nzs = d.nzl + 2 * halo + 1;
    if (const char* v = std::getenv("FPIC_DOMAIN_COMPACT"); v && std::strcmp(v, "0") == 0) return FPIC_OK;
    if (d.world < 2 || !st->own_fft || nzs >= st->nz) return FPIC_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    struct Arr { void** p; size_t per_node; };
    const Arr arrs[] = { { reinterpret_cast<void**>(&st->rho_fixed), sizeof(long long) }, { &st->rho, sizeof(T) }, { &st->phi, sizeof(T) }, { &st->E4, 4 * sizeof(T) },
                         { &st->Ey, 4 * sizeof(T) }, { &st->By, 4 * sizeof(T) }, { &st->B4n, 4 * sizeof(T) }, { reinterpret_cast<void**>(&st->Jfix), 3 * sizeof(long long) },
                         { &st->Bh[0], 4 * sizeof(T) }, { &st->Bh[1], 4 * sizeof(T) } }; // (Bh: the half-time arrays of a handle that ran the chained lattice step before)
    st->zs0 = ((d.z0 - halo) % st->nz + st->nz) % st->nz; // (held_plane() takes it in [0, nz))
    st->nzs = nzs;
    for (const Arr& a : arrs) {
        if (!*a.p) continue;
        HIP_TRY(h, hipFree(*a.p));
        *a.p = nullptr;
        h->bytes_grid -= st->nodes * a.per_node;
        if (int rc = dev_alloc(h, a.p, held_nodes(st) * a.per_node, &h->bytes_grid)) return rc;
    }
    if (st->hat) {
        HIP_TRY(h, hipFree(st->hat));
        st->hat = nullptr;
        h->bytes_grid -= hat_values<T>(st) * 2 * sizeof(T);
    }
    for (Species& s : st->sp) s.binned = false; // (nothing is binned yet: the rank's particles arrive after this)
    return FPIC_OK;
}

int domain_init(fpic_handle* h, int rank, int world, int ghost_planes, int migrate_every, int distributed_solve)
{
    State* st = h->es;
    if (st->dom) return fail(h, FPIC_ERR_STATE, "the handle is already decomposed");
    if (int rc0 = em_close_any(h)) return rc0;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, FPIC_ERR_INVALID_ARG, ".rank <- %d is outside a world of %d", rank, world);
    if (st->nz % world) return fail(h, FPIC_ERR_INVALID_ARG, ".world <- the %d planes along z do not divide into %d slabs", st->nz, world);
    const int nzl = st->nz / world;
    if (ghost_planes < 1 || ghost_planes >= nzl) return fail(h, FPIC_ERR_INVALID_ARG, ".ghost_planes <- must lie in [1, %d)", nzl);
    // two slabs are each other's lower AND upper neighbour: the G planes below and the G + 1 planes above a slab must be
    // different planes of the other one
    if (world == 2 && 2 * ghost_planes + 1 > nzl)
        return fail(h, FPIC_ERR_INVALID_ARG, ".ghost_planes <- with two slabs of %d planes at most %d ghost planes", nzl, (nzl - 1) / 2);
    if (migrate_every < 1) return fail(h, FPIC_ERR_INVALID_ARG, ".migrate_every <- must be at least 1");
    if (distributed_solve < 0 || distributed_solve > 2) return fail(h, FPIC_ERR_INVALID_ARG, ".distributed_solve <- %d is none of 0 (replicated), 1 (transposed spectrum), 2 (interface solve along z)", distributed_solve);
    if (st->solver == FPIC_SOLVER_YEE && world > 1 && 2 * (ghost_planes + 2) > nzl)
        return fail(h, FPIC_ERR_INVALID_ARG, ".ghost_planes <- the full-EM mode keeps ghost_planes + 2 halo planes per side: a slab of %d planes holds at most %d ghost planes",
                    nzl, nzl / 2 - 2);
    Domain* d = new (std::nothrow) Domain();
    if (!d) return fail(h, FPIC_ERR_OOM, "host allocation failed");
    st->dom = d;
    d->rank = rank; d->world = world; d->G = ghost_planes; d->nzl = nzl; d->z0 = rank * nzl; d->migrate_every = migrate_every;
    const size_t plane = static_cast<size_t>(st->nx) * st->ny;
    const size_t rec = h->prec == FPIC_F32 ? sizeof(MigRecord<float>) : sizeof(MigRecord<double>);
    size_t cap = 0;
    for (const Species& s : st->sp) cap = std::max(cap, s.cap);
    // records per migration message: a quarter of the largest species (a world of one never migrates)
    d->mig_cap = world == 1 ? 16u : static_cast<unsigned>(std::min<size_t>(std::max<size_t>(cap / 4, 4096), 0x7FFFFFFFu));
    uint64_t* acc = &h->bytes_grid;
    int rc;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&d->ghost_recv[0]), ghost_planes * plane * 8, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&d->ghost_recv[1]), (ghost_planes + 1) * plane * 8, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&d->counts_dev), kMigWords * sizeof(unsigned), acc)))
        return rc;
    for (int k = 0; k < 2; ++k)
        if ((rc = dev_alloc(h, &d->mig_send[k], d->mig_cap * rec, acc)) || (rc = dev_alloc(h, &d->mig_recv[k], d->mig_cap * rec, acc))) return rc;
    if (st->solver == FPIC_SOLVER_YEE) {
        d->H = ghost_planes + 2;
        for (int k = 0; k < 2; ++k)
            if ((rc = dev_alloc(h, reinterpret_cast<void**>(&d->j_recv[k]), static_cast<size_t>(d->H) * plane * 3 * sizeof(long long), acc))) return rc;
    }
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&d->counts_host), kMigWords * sizeof(unsigned)));
    if (const char* v = std::getenv("FPIC_DOMAIN_OVERLAP")) d->overlap = std::strcmp(v, "0") != 0;
    if (const char* v = std::getenv("FPIC_EM_CHAIN")) d->em_chain = std::strcmp(v, "0") != 0;
    if (const char* v = std::getenv("FPIC_TEST_FAULT")) d->test_fault = std::atoi(v);
    if (world > 1 && d->overlap) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIP_TRY(h, hipStreamCreateWithPriority(&d->comm_stream, hipStreamNonBlocking, hi)); // (its few workgroups must be placed while the push fills the chip)
        HIP_TRY(h, hipEventCreateWithFlags(&d->ev_boundary, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&d->ev_ghost, hipEventDisableTiming));
    }
    std::memset(d->counts_host, 0, kMigWords * sizeof(unsigned));
    for (Species& s : st->sp) s.n = 0; // the rank's particles arrive through domain_set_particles
    // the full-EM mode solves once, for the initial field: with the library's own transforms (whose decomposed solve is the
    // one handle's, bit for bit) its ranks can take the decomposed solve too, and then never hold the whole grid
    const bool yee_decomposed = st->solver == FPIC_SOLVER_YEE && st->own_fft && st->ny % world == 0;
    if (distributed_solve == 2 && world > 1 && st->solver == FPIC_SOLVER_YEE && !yee_decomposed)
        return fail(h, FPIC_ERR_INVALID_ARG, ".distributed_solve <- 2 on a full-EM handle needs a power-of-two grid whose %d rows along y divide into %d shares", st->ny, world);
    if (distributed_solve && world > 1 && (st->solver == FPIC_SOLVER_POISSON_FFT || yee_decomposed)) {
        d->phi_below = st->solver == FPIC_SOLVER_YEE ? d->H : ghost_planes + 1;
        d->phi_above = d->phi_below + 1;
        if (distributed_solve != 2 && st->ny % world) return fail(h, FPIC_ERR_INVALID_ARG, ".world <- the %d rows along y do not divide into %d shares for the decomposed solve", st->ny, world);
        if (ghost_planes + 2 > nzl) return fail(h, FPIC_ERR_INVALID_ARG, ".ghost_planes <- the decomposed solve needs ghost_planes + 2 <= %d planes per slab", nzl);
        d->distributed = true;
        d->nyl = st->ny / world;
        const size_t nxh = st->nx / 2 + 1, esz = h->esize;
        const size_t pitch = h->prec == FPIC_F32 ? row_pitch<float>(st) : row_pitch<double>(st);
        const size_t cbytes = pitch * st->ny * nzl * 2 * esz;
        if (distributed_solve == 2) { // the interface solve of fes_tri.hpp: the spectrum of the own planes stays where it is
            if (!st->own_fft) return fail(h, FPIC_ERR_INVALID_ARG, ".distributed_solve <- 2 (interface solve along z) runs on the library's own transforms: power-of-two grids");
            if (world > festri::kMaxRanks) return fail(h, FPIC_ERR_INVALID_ARG, ".distributed_solve <- 2 supports up to %d ranks (%d asked)", festri::kMaxRanks, world);
            if (st->nz > 1024) return fail(h, FPIC_ERR_INVALID_ARG, ".distributed_solve <- 2 holds the (0, 0) mode's line in one workgroup: nz <= 1024");
            d->interface_solve = true;
            d->tri_block = 2 * pitch * st->ny + nzl;
            if ((rc = dev_alloc(h, &d->hatA, cbytes, acc)) || (rc = dev_alloc(h, &d->tri, d->tri_block * world * 2 * esz, acc))) return rc;
        } else if ((rc = dev_alloc(h, &d->hatA, cbytes, acc)) || (rc = dev_alloc(h, &d->hatB, cbytes, acc)) || (rc = dev_alloc(h, &d->xbuf, cbytes, acc))) {
            return rc;
        }
        if (st->own_fft) { // the library's own passes work in place on hatA / hatB: no plans, no z-major copy
            // nothing on this rank reads or writes a node outside its slab, the ghost planes of the deposit (G below, G + 1
            // above) and the planes of phi their gradient needs (one more on each side): keep those
            if ((rc = h->prec == FPIC_F32 ? keep_slab_only<float>(h, ghost_planes + 2) : keep_slab_only<double>(h, ghost_planes + 2))) return rc;
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return FPIC_OK;
        }
        const fdyn::RocFFT& ff = fdyn::rocfft();
        if (!ff.ok) return fail(h, FPIC_ERR_STATE, "rocFFT is not available (%s)", ff.why.c_str());
        if ((rc = dev_alloc(h, &d->hatZ, cbytes, acc))) return rc;
        const rocfft_precision prec = h->prec == FPIC_F32 ? rocfft_precision_single : rocfft_precision_double;
        const size_t len2[2] = { static_cast<size_t>(st->nx), static_cast<size_t>(st->ny) };
        const size_t lenz[1] = { static_cast<size_t>(st->nz) };
        // the z pass runs on hatZ [nyl * nxh][nz]: nyl * nxh contiguous transforms of length nz
        const size_t zbatch = static_cast<size_t>(d->nyl) * nxh;
        if ((rc = fft_status(h, ff.plan_create(&d->p2f, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 2, len2, nzl, nullptr), "rocfft_plan_create (2-D forward)")) ||
            (rc = fft_status(h, ff.plan_create(&d->p2i, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 2, len2, nzl, nullptr), "rocfft_plan_create (2-D inverse)")) ||
            (rc = fft_status(h, ff.plan_create(&d->pzf, rocfft_placement_inplace, rocfft_transform_type_complex_forward, prec, 1, lenz, zbatch, nullptr), "rocfft_plan_create (z forward)")) ||
            (rc = fft_status(h, ff.plan_create(&d->pzi, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, prec, 1, lenz, zbatch, nullptr), "rocfft_plan_create (z inverse)")))
            return rc;
        rocfft_plan plans[4] = { d->p2f, d->p2i, d->pzf, d->pzi };
        rocfft_execution_info* infos[4] = { &d->i2f, &d->i2i, &d->izf, &d->izi };
        for (int k = 0; k < 4; ++k) {
            size_t wb = 0;
            if ((rc = fft_status(h, ff.execution_info_create(infos[k]), "rocfft_execution_info_create")) ||
                (rc = fft_status(h, ff.plan_get_work_buffer_size(plans[k], &wb), "rocfft_plan_get_work_buffer_size")))
                return rc;
            if (wb) {
                if ((rc = dev_alloc(h, &d->fft_work[k], wb, acc))) return rc;
                if ((rc = fft_status(h, ff.execution_info_set_work_buffer(*infos[k], d->fft_work[k], wb), "rocfft_execution_info_set_work_buffer"))) return rc;
            }
        }
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int domain_set_particles(fpic_handle* h, int species, uint64_t n, const void* pos_aos, const void* vel_aos, uint32_t first_id, int dtype)
{
    if (!h->es->dom) return fail(h, FPIC_ERR_STATE, "fpic_domain_set_particles needs fpic_domain_init first");
    if (int rc = check_species(h, species)) return rc;
    Species& s = h->es->sp[species];
    if (n > s.cap) return fail(h, FPIC_ERR_INVALID_ARG, ".position <- %llu particles exceed the species' capacity of %zu on this rank", static_cast<unsigned long long>(n), s.cap);
    if (!pos_aos || !vel_aos) return fail(h, FPIC_ERR_INVALID_ARG, ".position <- position and velocity are both needed");
    s.n = static_cast<size_t>(n);
    for (int k = 0; k < 2; ++k) { // slot order = upload order in both sets; the ids carry the caller's global index
        if (n) iota3_kernel<<<blocks_for(s.n), 256, 0, h->stream>>>(s.id[k], s.n, 0u);
    }
    HIP_TRY(h, hipGetLastError());
    s.ids_identity = true; // (both sets: slot = index, for the upload below)
    if (int rc = set_particles(h, species, pos_aos, vel_aos, 0, n, dtype)) return rc;
    if (n) iota3_kernel<<<blocks_for(s.n), 256, 0, h->stream>>>(s.id[s.cur], s.n, first_id);
    s.ids_identity = first_id == 0;
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

template <typename T, typename Out>
static int download_plain(fpic_handle* h, const Species& s, Out* host, int first)
{
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), s.n * 3 * sizeof(Out)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    get_plain3_kernel<T, Out><<<blocks_for(s.n), 256, 0, h->stream>>>(a + first * s.n_pad, a + (first + 1) * s.n_pad, a + (first + 2) * s.n_pad, s.n, stage);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, stage, s.n * 3 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

int domain_get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, uint32_t* ids, uint64_t capacity, uint64_t* n_out, int dtype)
{
    if (!h->es->dom) return fail(h, FPIC_ERR_STATE, "fpic_domain_get_particles needs a decomposed handle");
    if (int rc = check_species(h, species)) return rc;
    const Species& s = h->es->sp[species];
    if (n_out) *n_out = s.n;
    if (!pos_aos && !vel_aos && !ids) return FPIC_OK;
    if (capacity < s.n) return fail(h, FPIC_ERR_INVALID_ARG, ".capacity <- the rank holds %zu particles", s.n);
    if (dtype != FPIC_F32 && dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, ".dtype <- must be 0 (f32) or 1 (f64)");
    if (!s.n) return FPIC_OK;
    for (int pass = 0; pass < 2; ++pass) {
        void* dst = pass == 0 ? pos_aos : vel_aos;
        if (!dst) continue;
        int rc;
        if (h->prec == FPIC_F32)
            rc = dtype == FPIC_F32 ? download_plain<float, float>(h, s, static_cast<float*>(dst), 3 * pass) : download_plain<float, double>(h, s, static_cast<double*>(dst), 3 * pass);
        else
            rc = dtype == FPIC_F32 ? download_plain<double, float>(h, s, static_cast<float*>(dst), 3 * pass) : download_plain<double, double>(h, s, static_cast<double*>(dst), 3 * pass);
        if (rc) return rc;
    }
    if (ids) {
        HIP_TRY(h, hipMemcpyAsync(ids, s.id[s.cur], s.n * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return FPIC_OK;
}

int domain_stats(fpic_handle* h, uint64_t* migrated, uint64_t* lost)
{
    if (!h->es->dom) return fail(h, FPIC_ERR_STATE, "the handle is not decomposed");
    if (migrated) *migrated = h->es->dom->migrated;
    if (lost) *lost = h->es->dom->lost;
    return FPIC_OK;
}

namespace {

// One queue for the whole group while it runs (the copies between members then need no further ordering): every
// member's own stream — its own, or the caller's (fpic_set_stream) — is drained, is swapped for rank 0's,
// and comes back afterwards, ordered behind what the group queued.
struct GroupStreams {
    std::vector<fpic_handle*> hs;
    std::vector<hipStream_t> saved;
    hipStream_t shared = nullptr;
    int enter(fpic_handle** handles, int n)
    {
        shared = handles[0]->stream;
        for (int r = 0; r < n; ++r) {
            fpic_handle* h = handles[r];
            if (h->stream != shared) HIP_TRY(h, hipStreamSynchronize(h->stream)); // work queued before the group call
            hs.push_back(h);
            saved.push_back(h->stream);
            h->stream = shared;
        }
        return FPIC_OK;
    }
    ~GroupStreams()
    {
        hipEvent_t done = nullptr;
        bool other = false;
        for (size_t r = 0; r < hs.size(); ++r) other |= saved[r] != shared;
        if (other && hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(done, shared);
        for (size_t r = 0; r < hs.size(); ++r) {
            hs[r]->stream = saved[r];
            if (saved[r] != shared) {
                if (done) (void)hipStreamWaitEvent(saved[r], done, 0);
                else (void)hipStreamSynchronize(shared);
            }
        }
        if (done) (void)hipEventDestroy(done);
    }
};

} // namespace

int group_run(fpic_handle** hs, int n, int what, int ncalls)
{
    Ranks rk;
    rk.hs.assign(hs, hs + n);
    rk.rccl = false;
    fpic_handle* h0 = hs[0];
    for (int r = 0; r < n; ++r) {
        fpic_handle* h = hs[r];
        if (!h || !h->es || !h->es->dom) return fail(h0, FPIC_ERR_INVALID_ARG, ".handles <- member %d is not a decomposed CART3D handle", r);
        const Domain& d = *h->es->dom;
        if (d.world != n || d.rank != r) return fail(h0, FPIC_ERR_INVALID_ARG, ".handles <- member %d is rank %d of %d, the group has %d members", r, d.rank, d.world, n);
        if (h->prec != h0->prec || h->es->nodes != h0->es->nodes || h->device != h0->device || h->es->sp.size() != h0->es->sp.size())
            return fail(h0, FPIC_ERR_INVALID_ARG, ".handles <- member %d differs in precision, grid, device or species", r);
        if (h->comm) return fail(h0, FPIC_ERR_INVALID_ARG, ".handles <- member %d has a communicator; a group is the in-process exchange", r);
        if (what == 1 && !h->es->fields_ready) return fail(h0, FPIC_ERR_STATE, "step() before precalc()");
    }
    GroupStreams streams;
    if (int e = streams.enter(hs, n)) return e;
    // a failure inside the group is recorded on the member it happened on: the group reports through its first member
    auto report = [&](int rc) {
        if (rc != FPIC_OK && h0->err.empty())
            for (int r = 1; r < n; ++r)
                if (!hs[r]->err.empty()) { h0->err = hs[r]->err; break; }
        return rc;
    };
    for (int r = 0; r < n; ++r) hs[r]->err.clear();
    if (what == 0) return report(h0->prec == FPIC_F32 ? dom_precalc<float>(rk) : dom_precalc<double>(rk));
    if (what == 2) return report(h0->es->solver != FPIC_SOLVER_YEE ? FPIC_OK : (h0->prec == FPIC_F32 ? dom_density<float>(rk) : dom_density<double>(rk)));
    for (int k = 0; k < 2 * ncalls; ++k)
        if (int rc = h0->prec == FPIC_F32 ? dom_substep<float>(rk) : dom_substep<double>(rk)) return report(rc);
    return FPIC_OK;
}

} // namespace fes
