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

struct Species {
    double mass = 0, charge = 0;
    int Z = 1;
    size_t n = 0;     // particles held now (a decomposed run gains and loses particles by migration)
    size_t cap = 0;   // capacity of the arrays
    size_t n_pad = 0;
    void* slab[2] = {};
    uint32_t* id[2] = {};
    int cur = 0;
    // two bin tables: [wl] describes the live particle order, [wl ^ 1] is laid out by the next binning
    // (which may be the next push, see rebin_pending)
    uint32_t *tile_count = nullptr, *tile_cursor = nullptr;
    // a migration rides on the next re-binning push: arrivals appended at [tail_first, tail_first + tail_count) of the
    // current set, n_after = the population once that push has compacted the set
    size_t tail_first = 0, tail_count = 0, n_after = 0;
    bool ids_identity = true;   // slot s still holds the caller's particle s (no binning yet): uploads go straight to their slots
    uint32_t *tile_start2[2] = {}, *nwork2[2] = {};
    BlockWork* work2[2] = {};
    int wl = 0;
    size_t work_cap = 0;
    bool binned = false;
    bool census_fresh = false;  // tile_count holds the census of the current positions (written by the last push)
    bool rebin_pending = false; // tables [wl ^ 1] are laid out from that census: the next push re-bins
    bool rebin_now = false;     // the push in flight is that re-binning (a push in two parts decides once)
    uint32_t* chunk_census = nullptr;  // 27 words per work item: the new positions of the last in-place launch by neighbour slot
    bool chunk_census_fresh = false;   // ... of the live work list and slots: the next re-binning launch need not count
    int chunk_census_form = 0;         // ... written by a whole launch (0) or by the two parts of a rank's launch (1),
                                       //     bit 1: over the joint work list of every species (State::joint_work) instead of its own
    uint64_t chunk_census_list = 0;    // ... and which build of the joint list its items are those of (State::joint_build)
    size_t chunk_census_items = 0;     // work items the census has room for
    uint64_t layout = 0;               // counts the changes of the live bin table (what a joint work list is built from)
    void* em_args = nullptr;           // EmPushArgs of the last full-EM launch, resident for the kernel's out-of-line paths
};

struct State {
    int nx = 0, ny = 0, nz = 0;
    double lx = 0, ly = 0, lz = 0, W = 1;
    size_t nodes = 0;
    // the planes the node arrays hold (fes_kernels.hpp, Held): all nz of them, or — a rank of a compact decomposition —
    // the slab with its halo: zs0 = z0 - H, nzs = nzl + 2 H + 1 planes
    int zs0 = 0, nzs = 0;
    int solver = FPIC_SOLVER_NONE;
    int ltx = 4, lty = 4, ltz = 3; // log2 of the tile edges: 16x16x8 cells (electrostatic), 8x8x8 (full EM)
    int ntx = 0, nty = 0, ntz = 0;
    uint32_t ntiles = 0; // + 1 always-empty bin (the scan kernel's clipped bin)
    long long* rho_fixed = nullptr;
    void *rho = nullptr, *hat = nullptr, *phi = nullptr, *E4 = nullptr;
    // full EM (solver = YEE): the lattice's E and B, the node-centred B (E4 holds the node-centred E), the integer current grid
    void *Ey = nullptr, *By = nullptr, *B4n = nullptr;
    // the chained lattice step of an undecomposed full-EM handle (em_chain_kernel): B at half time, two arrays taken in turns;
    // em_open: Ey is E of the integer time reached, Bh[bh_cur] is B half a step before it, By is stale until em_close()
    void* Bh[2] = { nullptr, nullptr };
    int bh_cur = 0;
    bool em_open = false;
    long long* Jfix = nullptr;
    double* k2[3] = {};
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info_f = nullptr, info_i = nullptr;
    void *work_f = nullptr, *work_i = nullptr;
    double B0[3] = { 0, 0, 0 };
    unsigned long long* spilled = nullptr;
    // the work list of a launch that pushes every binned species (Push3Joint): items (tile, k), rebuilt when a species'
    // bin table has changed
    fpic::BlockWork* joint_work = nullptr;
    uint32_t* joint_nwork = nullptr;
    size_t joint_cap = 0;
    std::vector<std::pair<size_t, uint64_t>> joint_built_from; // (species, layout) of the list in joint_work
    uint64_t joint_build = 0;          // counts the rebuilds (a per-item census belongs to the list it was written over)
    bool joint_now = false;            // the two parts of one sub-step's launch use the same list
    unsigned long long* spilled_host = nullptr; // pinned, 2 lagged slots
    hipEvent_t spill_event[2] = {};
    bool spill_pending[2] = {};
    unsigned long long spill_seq = 0, last_spill = 0;
    int substeps_since_bin = 0;
    bool fields_ready = false;
    // power-of-two grids: the Poisson solve runs on the library's own FFT passes (fes_fft.hpp), which read the integer
    // charge grid directly; rho (T) is then formed only when somebody reads it
    bool own_fft = false, rho_fresh = true;
    void* fft_tw[3] = {};   // twiddle tables exp(-2 pi i t / n) of the three axes (T pairs)
    std::vector<Species> sp;
    struct Domain* dom = nullptr; // z-slab decomposition over several GPUs (fpic_domain_init)
};

// Spatial decomposition: rank r of `world` owns the particles whose cell lies in the
// planes [z0, z0 + nzl) and G ghost planes on either side, in which its particles may still sit and deposit
// until the next migration.  Per sub-step: ghost-plane reduce of the int64 charge grid with the two
// neighbours (exact), all-gather of the owned planes of rho, the field solve on every rank; every
// `migrate_every` sub-steps the particles that left the slab move to the neighbour that owns them.
struct Domain {
    int rank = 0, world = 1, G = 2, nzl = 0, z0 = 0;
    int migrate_every = 4;
    int substeps_since_migration = 0;
    long long* ghost_recv[2] = {};      // [0]: from the slab above (its lower ghost planes, G), [1]: from below (G + 1)
    void* mig_send[2] = {};             // [0]: to the slab below, [1]: to the slab above
    void* mig_recv[2] = {};             // [0]: from above, [1]: from below
    unsigned mig_cap = 0;               // records per buffer
    // full EM: halo planes of the lattice fields / ghost planes of the current on each side (G + 2), and where the
    // neighbours' current ghost planes arrive (3 int64 per node)
    int H = 0;
    long long* j_recv[2] = {};
    bool halos_stale = false;           // lattice fields restored from a checkpoint: the halo planes are refreshed before the next sub-step
    // per species a block of 8 words: down, up, lost, overflow | received from above, from below | -, - ; after the
    // kMigSpecies blocks one more, whose first word is the ranks' agreement (agree_max)
    unsigned* counts_dev = nullptr;
    unsigned* counts_host = nullptr;    // pinned copy
    int mig_sp = 0;                     // the species whose payload the exchange X_MIG_PAYLOAD moves
    // the ghost-plane exchange of a sub-step runs on a stream of its own while the interior of the slab is pushed
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_boundary = nullptr, ev_ghost = nullptr;
    bool overlap = true;                // FPIC_DOMAIN_OVERLAP=0: everything on the handle's stream, one launch per species
    bool em_chain = true, em_chain_agreed = false; // FPIC_EM_CHAIN=0 (read by fpic_domain_init, agreed by the ranks before the first full-EM sub-step)
    // TEST SWITCH (FPIC_TEST_FAULT, read by fpic_domain_init; tests/test_gpu_fake_rccl.py's negative controls): bit 0 drops the
    // wait of the communicator's stream for the handle's stream (comm_fork), bit 1 the wait of the handle's stream for the
    // exchange (comm_join) — the two dependencies a stream-ordered transport must show as wrong bits when they are missing
    int test_fault = 0;
    uint64_t migrated = 0, lost = 0, deferred = 0; // deferred: leavers that did not fit a message and left with a later one
    // slab-decomposed Poisson solve (distributed = true): 2-D transforms of the owned planes, transpose over the ranks,
    // transforms along z of the rank's share of the ky rows, and back; otherwise every rank transforms the whole grid
    bool distributed = false;
    int nyl = 0;
    int phi_below = 0, phi_above = 0;   // planes of the potential a rank receives from its neighbours after the decomposed solve
    void *hatA = nullptr, *hatB = nullptr, *xbuf = nullptr; // [nzl][ny][nxh], [nz][nyl][nxh], transposition staging (complex T each)
    rocfft_plan p2f = nullptr, p2i = nullptr, pzf = nullptr, pzi = nullptr;
    rocfft_execution_info i2f = nullptr, i2i = nullptr, izf = nullptr, izi = nullptr;
    void* fft_work[4] = {};
    void* hatZ = nullptr;               // hatB turned to [nyl * nxh][nz]: the z pass is contiguous there
    // distributed_solve = 2 (fes_tri.hpp): no transposition — the decomposed direction is a periodic tridiagonal system per
    // (kx, ky) mode, reduced per rank to two interface planes; tri = [world][2 planes of the half spectrum + nzl values of
    // the (0, 0) mode's line] (complex T), all-gathered in place; tri_block = complex values per rank
    bool interface_solve = false;
    void* tri = nullptr;
    size_t tri_block = 0;
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

Held held_of(const State* st) { return Held{ st->zs0, st->nzs }; }
bool compact(const State* st) { return st->nzs != st->nz; }
// place of global plane k (any integer: periodic) in the node arrays; the caller names held planes only
size_t lp(const State* st, int k)
{
    const int l = (((k - st->zs0) % st->nz) + st->nz) % st->nz;
    return static_cast<size_t>(l);
}
size_t held_nodes(const State* st) { return static_cast<size_t>(st->nx) * st->ny * st->nzs; }
// zero `count` planes from global plane `first` on (periodic) of a node array with `per_plane` bytes per plane: one run
// of the array, or two where the planes wrap past its end
int zero_planes(fpic_handle* h, void* base, size_t per_plane, int first, int count)
{
    const State* st = h->es;
    count = std::min(count, st->nzs);
    const size_t l0 = lp(st, first);
    const size_t head = std::min<size_t>(count, static_cast<size_t>(st->nzs) - l0);
    HIP_TRY(h, hipMemsetAsync(static_cast<char*>(base) + l0 * per_plane, 0, head * per_plane, h->stream));
    if (static_cast<size_t>(count) > head) HIP_TRY(h, hipMemsetAsync(base, 0, (count - head) * per_plane, h->stream));
    return FPIC_OK;
}

template <typename T>
int alloc_species(fpic_handle* h, Species& s)
{
    State* st = h->es;
    if (s.cap < s.n) s.cap = s.n;
    s.n_pad = (s.cap + 1023) / 1024 * 1024;
    for (int k = 0; k < 2; ++k) {
        if (int rc = dev_alloc(h, &s.slab[k], 6 * s.n_pad * sizeof(T), &h->bytes_particles)) return rc;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&s.id[k]), s.n_pad * sizeof(uint32_t), &h->bytes_particles)) return rc;
        init3_kernel<T><<<blocks_for(s.n_pad), 256, 0, h->stream>>>(static_cast<T*>(s.slab[k]), s.n_pad, s.id[k]);
        HIP_TRY(h, hipGetLastError());
    }
    s.work_cap = (s.cap + kChunk3 - 1) / kChunk3 + st->ntiles;
    uint64_t* acc = &h->bytes_grid;
    int rc;
    s.chunk_census_items = s.work_cap;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&s.chunk_census), sizeof(uint32_t) * kNbr3 * s.work_cap, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&s.tile_count), sizeof(uint32_t) * st->ntiles, acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&s.tile_cursor), sizeof(uint32_t) * (st->ntiles + fpic::kSortMaxBins + 1), acc))) // + chunk_first of the two-level binning
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
    if (s.chunk_census) (void)hipFree(s.chunk_census);
    if (s.em_args) (void)hipFree(s.em_args);
    for (void* p : { static_cast<void*>(s.tile_count), static_cast<void*>(s.tile_cursor), static_cast<void*>(s.tile_start2[0]),
                     static_cast<void*>(s.tile_start2[1]), static_cast<void*>(s.nwork2[0]), static_cast<void*>(s.nwork2[1]),
                     static_cast<void*>(s.work2[0]), static_cast<void*>(s.work2[1]) })
        if (p) (void)hipFree(p);
}

bool interior_layers(const State* st, uint32_t& lo, uint32_t& hi);

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
    a.held = held_of(st);
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
    a.tile_start = s.tile_start2[s.wl];
    a.chunk_census = nullptr;
    a.part = 0; a.tiles_per_layer = static_cast<uint32_t>(st->ntx) * st->nty; a.layer_lo = a.layer_hi = 0;
    a.spilled = st->spilled;
    a.tile_count = s.tile_count;
    a.id = s.id[s.cur];
    a.dst_slab = static_cast<T*>(s.slab[s.cur ^ 1]);
    a.dst_id = s.id[s.cur ^ 1];
    a.dst_tile_start = s.tile_start2[s.wl ^ 1];
    a.dst_tile_cursor = s.tile_cursor;
    return a;
}

// The joint work list of `set` (binned species): items (tile, k), k-th piece of kChunk3 slots of the tile in every species'
// bin table.  Rebuilt when a member's table has changed since it was built.
template <typename T>
int ensure_joint_list(fpic_handle* h, const std::vector<size_t>& set)
{
    State* st = h->es;
    std::vector<std::pair<size_t, uint64_t>> sig;
    size_t need = st->ntiles + 1;
    for (size_t i : set) { sig.push_back({ i, st->sp[i].layout }); need += st->sp[i].cap / kChunk3 + 1; }
    if (need > st->joint_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (st->joint_work) (void)hipFree(st->joint_work);
        st->joint_work = nullptr;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&st->joint_work), sizeof(BlockWork) * need, &h->bytes_grid)) return rc;
        if (!st->joint_nwork)
            if (int rc = dev_alloc(h, reinterpret_cast<void**>(&st->joint_nwork), sizeof(uint32_t), &h->bytes_grid)) return rc;
        st->joint_cap = need;
        st->joint_built_from.clear();
    }
    for (size_t i : set) { // the per-item census of every member has room for the joint list's items
        Species& s = st->sp[i];
        if (s.chunk_census_items >= st->joint_cap) continue;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (s.chunk_census) (void)hipFree(s.chunk_census);
        s.chunk_census = nullptr;
        if (int rc = dev_alloc(h, reinterpret_cast<void**>(&s.chunk_census), sizeof(uint32_t) * kNbr3 * st->joint_cap, &h->bytes_grid)) return rc;
        s.chunk_census_items = st->joint_cap;
        s.chunk_census_fresh = false;
    }
    if (sig == st->joint_built_from) return FPIC_OK;
    JointTables tabs{};
    tabs.n = static_cast<int>(set.size());
    for (size_t k = 0; k < set.size(); ++k) tabs.tile_start[k] = st->sp[set[k]].tile_start2[st->sp[set[k]].wl];
    joint_scan_kernel<<<1, 1024, 0, h->stream>>>(tabs, st->ntiles, static_cast<uint32_t>(kChunk3), st->joint_work, st->joint_nwork);
    HIP_TRY(h, hipGetLastError());
    st->joint_built_from = sig;
    st->joint_build++;
    return FPIC_OK;
}

// The push (or, DEPOSIT_ONLY, the deposit) of every species of the handle.  part 0: whole; a rank of a decomposition may
// push in two parts: 1 = the tile layers along the slab's faces (and the arrivals of a migration), 2 = the interior; the
// re-binning decision, the census reset and the switch of the particle sets are taken once.  Binned species share ONE
// launch where they can (two or more of them, all re-binning or none): a tile's window is then staged and flushed once
// for all of them (FPIC_PUSH_JOINT=0: one launch per species, a development switch).
template <typename T, bool DEPOSIT_ONLY>
int launch_push_all(fpic_handle* h, int part = 0)
{
    State* st = h->es;
    const bool has_b = st->B0[0] != 0 || st->B0[1] != 0 || st->B0[2] != 0;
    std::vector<size_t> tiled;
    for (size_t i = 0; i < st->sp.size(); ++i) {
        Species& s = st->sp[i];
        if (s.n == 0) continue;
        if (s.binned && st->solver != FPIC_SOLVER_YEE) { tiled.push_back(i); continue; }
        if constexpr (DEPOSIT_ONLY) {
            // the full-EM mode bins by 8x8x8-cell tiles: its charge grid (density(), the start field) has a tiled form of its own
            if (s.binned) {
                EmTileArgs<T> t{};
                t.p.slab = static_cast<T*>(s.slab[s.cur]); t.p.stride = s.n_pad; t.p.n = s.n;
                t.p.nx = st->nx; t.p.ny = st->ny; t.p.nz = st->nz;
                t.p.held = held_of(st);
                t.p.Z = s.Z;
                t.ntx = st->ntx; t.nty = st->nty; t.ntz = st->ntz;
                t.work = s.work2[s.wl]; t.nwork = s.nwork2[s.wl];
                t.part = part; t.tiles_per_layer = static_cast<uint32_t>(st->ntx) * st->nty;
                interior_layers(st, t.layer_lo, t.layer_hi);
                t.spilled = st->spilled;
                t.tile_start = s.tile_start2[s.wl];
                em_rho_tiles_kernel<T><<<static_cast<unsigned>(s.work_cap), kEmRhoThreads, 0, h->stream>>>(t, reinterpret_cast<unsigned long long*>(st->rho_fixed));
                HIP_TRY(h, hipGetLastError());
                continue;
            }
        }
        if (part == 2) continue; // (an unbinned species is pushed whole with the first part)
        Push3Args<T> a = push_args<T>(h, s);
        const size_t lanes = (s.n + Vec16<T>::N - 1) / Vec16<T>::N;
        if (has_b && !DEPOSIT_ONLY) push3_flat_kernel<T, true, DEPOSIT_ONLY><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
        else push3_flat_kernel<T, false, DEPOSIT_ONLY><<<blocks_for(lanes), 256, 0, h->stream>>>(a);
        HIP_TRY(h, hipGetLastError());
        if (!DEPOSIT_ONLY) s.census_fresh = s.chunk_census_fresh = false;
    }
    if (tiled.empty()) return FPIC_OK;
    constexpr size_t lds = push3_lds_bytes<T>();
    // one launch for all of them?  (decided with the first part; the second part walks the same list)
    if (part != 2) {
        const char* v = std::getenv("FPIC_PUSH_JOINT");
        const bool allowed = !(v && std::strcmp(v, "0") == 0);
        bool same = true;
        for (size_t i : tiled) same &= DEPOSIT_ONLY || st->sp[i].rebin_pending == st->sp[tiled[0]].rebin_pending;
        st->joint_now = allowed && same && tiled.size() >= 2 && tiled.size() <= static_cast<size_t>(kJointMax);
    }
    const bool joint = st->joint_now && tiled.size() >= 2;
    if (joint)
        if (int rc = ensure_joint_list<T>(h, tiled)) return rc;
    // per species: the state a launch starts from, and its arguments
    std::vector<Push3Args<T>> args(tiled.size());
    bool rebin = false;
    for (size_t k = 0; k < tiled.size(); ++k) {
        Species& s = st->sp[tiled[k]];
        Push3Args<T>& a = args[k];
        a = push_args<T>(h, s);
        if constexpr (!DEPOSIT_ONLY) {
            if (part != 2) {
                s.rebin_now = s.rebin_pending;
                s.census_fresh = s.rebin_pending = false;
                HIP_TRY(h, hipMemsetAsync(s.tile_count, 0, sizeof(uint32_t) * st->ntiles, h->stream));
            }
            a.part = part;
            interior_layers(st, a.layer_lo, a.layer_hi);
            // An in-place launch leaves the per-item census the next re-binning launch starts from.  On a rank of a
            // decomposition a migration lies in between: it changes slots of the layers along the faces only (leavers; the
            // arrivals sit in the tail), so those items count again and the interior's read theirs.  The census belongs
            // to the launch form that wrote it: its items are those of one work list (the species' own or the joint one),
            // and a two-part launch hands one straddling group of slots to another item than a whole launch does
            // (species_groups).
            const bool ranks = st->dom && st->dom->world > 1;
            const int form = (part == 0 ? 0 : 1) | (joint ? 2 : 0);
            const uint64_t list = joint ? st->joint_build : 0;
            a.chunk_census = !s.rebin_now || (s.chunk_census_fresh && s.chunk_census_form == form && s.chunk_census_list == list) ? s.chunk_census : nullptr;
            a.census_interior_only = ranks ? 1 : 0;
            if (part != 1) s.chunk_census_fresh = !s.rebin_now; // (after the last part)
            if (!s.rebin_now) { s.chunk_census_form = form; s.chunk_census_list = list; }
            if (k == 0) rebin = s.rebin_now;
        }
    }
    // the launches: one over the joint list, or one per species over its own
    auto launch = [&](const Push3Joint<T>& J, unsigned grid, bool re) -> int {
        if constexpr (DEPOSIT_ONLY) {
            push3_tiles_kernel<T, false, true><<<grid, kPushThreads3, lds, h->stream>>>(J);
        } else {
            if (re && has_b) push3_tiles_kernel<T, true, false, true><<<grid, kPushThreads3, lds, h->stream>>>(J);
            else if (re) push3_tiles_kernel<T, false, false, true><<<grid, kPushThreads3, lds, h->stream>>>(J);
            else if (has_b) push3_tiles_kernel<T, true, false><<<grid, kPushThreads3, lds, h->stream>>>(J);
            else push3_tiles_kernel<T, false, false><<<grid, kPushThreads3, lds, h->stream>>>(J);
        }
        HIP_TRY(h, hipGetLastError());
        return FPIC_OK;
    };
    if (joint) {
        Push3Joint<T> J{};
        J.nsp = static_cast<int>(tiled.size());
        for (size_t k = 0; k < tiled.size(); ++k) J.sp[k] = args[k];
        J.work = st->joint_work; J.nwork = st->joint_nwork; J.chunk = static_cast<uint32_t>(kChunk3);
        if (int rc = launch(J, static_cast<unsigned>(st->joint_cap), rebin)) return rc;
    } else {
        for (size_t k = 0; k < tiled.size(); ++k) {
            Species& s = st->sp[tiled[k]];
            Push3Joint<T> J{};
            J.nsp = 1; J.sp[0] = args[k];
            J.work = args[k].work; J.nwork = args[k].nwork; J.chunk = 0;
            if (int rc = launch(J, static_cast<unsigned>(s.work_cap), DEPOSIT_ONLY ? false : s.rebin_now)) return rc;
        }
    }
    if constexpr (!DEPOSIT_ONLY) {
        for (size_t k = 0; k < tiled.size(); ++k) {
            Species& s = st->sp[tiled[k]];
            const bool re = s.rebin_now;
            if (part != 2 && re && s.tail_count) { // the arrivals of the migration that asked for this re-binning
                if (has_b) push3_tail_kernel<T, true><<<blocks_for(s.tail_count), 256, 0, h->stream>>>(args[k], s.tail_first, s.tail_count);
                else push3_tail_kernel<T, false><<<blocks_for(s.tail_count), 256, 0, h->stream>>>(args[k], s.tail_first, s.tail_count);
                HIP_TRY(h, hipGetLastError());
            }
            if (part != 1) {
                s.census_fresh = true;
                if (re) { // this launch was the binning: the other set and the other tables are live now
                    s.cur ^= 1;
                    s.wl ^= 1;
                    s.layout++;
                    if (s.n_after) s.n = s.n_after;
                    s.tail_first = s.tail_count = s.n_after = 0;
                }
                s.rebin_now = false;
            }
        }
    }
    return FPIC_OK;
}

// The binning of a grid with more tiles than an LDS histogram holds (512^3: 65 536 tiles of 16 x 16 x 8, 262 144 of 8^3): census
// and scatter both in two levels, no global atomic per particle anywhere (fes_kernels.hpp, bin3_count_coarse_kernel):
// coarse census, coarse scan, coarse scatter; tile census of the coarse-sorted array, tile scan (+ work list), tile scatter.
// Same result as the one-level census (FPIC_CENSUS_GLOBAL=1 keeps that form): the bin table is a function of the positions.
template <typename T>
int launch_bin_two_level_census(fpic_handle* h, Species& s)
{
    State* st = h->es;
    const bool em = st->ltx == kEL;
    const int nw = s.wl ^ 1;
    uint32_t div = 1;
    while (div * div < st->ntiles) ++div;
    const uint32_t ncoarse = (st->ntiles + div - 1) / div;
    uint32_t* aux = s.tile_cursor + st->ntiles;   // [ncoarse + 1]: the coarse counts and the live total, then chunk_first of the tile pass
    auto columns = [&](int from) {
        fpic::SortColumns<T, 6, false> c{};
        for (int f = 0; f < 6; ++f) {
            c.src[f] = static_cast<const T*>(s.slab[from]) + f * s.n_pad;
            c.dst[f] = static_cast<T*>(s.slab[from ^ 1]) + f * s.n_pad;
        }
        c.src_id = s.id[from]; c.dst_id = s.id[from ^ 1];
        return c;
    };
    const size_t lds = fpic::sort_scatter_lds(sizeof(T));
    const unsigned nc = blocks_for(s.n, fpic::kSortChunk), ncount = blocks_for(s.n, 1024 * kCoarsePer);
    auto run = [&](auto key) -> int {
        using Key = decltype(key);
        auto kern = fpic::sort_scatter_kernel<T, 6, false, Key>;
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        HIP_TRY(h, hipMemsetAsync(s.tile_cursor, 0, sizeof(uint32_t) * (st->ntiles + ncoarse + 1), h->stream));
        bin3_count_coarse_kernel<T, Key><<<ncount, 1024, 0, h->stream>>>(static_cast<const T*>(s.slab[s.cur]), s.n_pad, s.n, key, div, ncoarse, aux);
        coarse_scan_kernel<<<1, 1024, 0, h->stream>>>(aux, ncoarse, div, s.tile_start2[nw]);
        kern<<<nc, fpic::kSortThreads, lds, h->stream>>>(columns(s.cur), s.n, key, st->ntiles, div, ncoarse, s.tile_start2[nw], s.tile_cursor, nullptr);
        bin3_count_sorted_kernel<T, Key><<<ncount, 1024, 0, h->stream>>>(static_cast<const T*>(s.slab[s.cur ^ 1]), s.n_pad, aux + ncoarse, key, div, s.tile_count);
        bin_scan_kernel<<<1, 1024, 0, h->stream>>>(s.tile_count, st->ntiles, s.tile_start2[nw], s.tile_cursor, s.work2[nw], s.nwork2[nw], static_cast<uint32_t>(kChunk3));
        HIP_TRY(h, hipMemsetAsync(s.tile_cursor, 0, sizeof(uint32_t) * st->ntiles, h->stream));
        fpic::sort_chunks_kernel<<<1, 1024, 0, h->stream>>>(s.tile_start2[nw], st->ntiles, div, ncoarse, aux);
        kern<<<nc + ncoarse, fpic::kSortThreads, lds, h->stream>>>(columns(s.cur ^ 1), s.n, key, st->ntiles, div, ncoarse, s.tile_start2[nw], s.tile_cursor, aux);
        return FPIC_OK;
    };
    if (int rc = em ? run(BoxTileKey<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>{ st->nx, st->ny, st->nz, st->ntx, st->nty }) : run(BoxTileKey<T>{ st->nx, st->ny, st->nz, st->ntx, st->nty }))
        return rc;
    HIP_TRY(h, hipGetLastError());
    s.wl = nw;                // (two passes: the compact sorted array is back in the set it started in)
    s.layout++;
    s.binned = true;
    s.ids_identity = false;
    s.census_fresh = s.rebin_pending = s.chunk_census_fresh = false;
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
    const bool em = st->ltx == kEL;
    const bool many_tiles = st->ntiles > static_cast<uint32_t>(kMaxTiles3); // no LDS histogram of that size
    if (many_tiles && !std::getenv("FPIC_CENSUS_GLOBAL")) return launch_bin_two_level_census<T>(h, s);
    if (many_tiles && em) bin3_count_global_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ><<<blocks_for(s.n), 256, 0, h->stream>>>(src, s.n_pad, s.n, st->nx, st->ny, st->nz, st->ntx, st->nty, s.tile_count);
    else if (many_tiles) bin3_count_global_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(src, s.n_pad, s.n, st->nx, st->ny, st->nz, st->ntx, st->nty, s.tile_count);
    else if (em) bin3_count_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ><<<nb, 256, shmem, h->stream>>>(src, s.n_pad, s.n, st->nx, st->ny, st->nz, st->ntx, st->nty, st->ntiles, s.tile_count);
    else bin3_count_kernel<T><<<nb, 256, shmem, h->stream>>>(src, s.n_pad, s.n, st->nx, st->ny, st->nz, st->ntx, st->nty, st->ntiles, s.tile_count);
    bin_scan_kernel<<<1, 1024, 0, h->stream>>>(s.tile_count, st->ntiles, s.tile_start2[nw], s.tile_cursor, s.work2[nw], s.nwork2[nw], static_cast<uint32_t>(kChunk3));
    // large populations: scatter staged through LDS (fpic_kernels.hpp, sort_scatter_kernel), in two levels when there
    // are many tiles; after two passes the compact sorted array is back in the set it started in
    const bool staged = s.n >= h->two_level_min || many_tiles;
    bool two_level = false;
    if (staged) {
        uint32_t div = 1; // few tiles: one staged pass is enough
        while (st->ntiles > 64 && div * div < st->ntiles) ++div;
        two_level = div > 1;
        const uint32_t ncoarse = (st->ntiles + div - 1) / div;
        uint32_t* chunk_first = s.tile_cursor + st->ntiles;
        auto columns = [&](int from) {
            fpic::SortColumns<T, 6, false> c{};
            for (int f = 0; f < 6; ++f) {
                c.src[f] = static_cast<const T*>(s.slab[from]) + f * s.n_pad;
                c.dst[f] = static_cast<T*>(s.slab[from ^ 1]) + f * s.n_pad;
            }
            c.src_id = s.id[from]; c.dst_id = s.id[from ^ 1];
            return c;
        };
        const size_t lds = fpic::sort_scatter_lds(sizeof(T));
        const unsigned nc = blocks_for(s.n, fpic::kSortChunk);
        auto run = [&](auto key) -> int {
            auto kern = fpic::sort_scatter_kernel<T, 6, false, decltype(key)>;
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            fpic::sort_chunks_kernel<<<1, 1024, 0, h->stream>>>(s.tile_start2[nw], st->ntiles, div, ncoarse, chunk_first);
            kern<<<nc, fpic::kSortThreads, lds, h->stream>>>(columns(s.cur), s.n, key, st->ntiles, div, ncoarse, s.tile_start2[nw], s.tile_cursor, nullptr);
            if (two_level) {
                HIP_TRY(h, hipMemsetAsync(s.tile_cursor, 0, sizeof(uint32_t) * st->ntiles, h->stream));
                kern<<<nc + ncoarse, fpic::kSortThreads, lds, h->stream>>>(columns(s.cur ^ 1), s.n, key, st->ntiles, div, ncoarse, s.tile_start2[nw],
                                                                          s.tile_cursor, chunk_first);
            }
            return FPIC_OK;
        };
        if (int rc = em ? run(BoxTileKey<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>{ st->nx, st->ny, st->nz, st->ntx, st->nty })
                        : run(BoxTileKey<T>{ st->nx, st->ny, st->nz, st->ntx, st->nty }))
            return rc;
    } else if (em) {
        bin3_scatter_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ><<<nb, 256, shmem, h->stream>>>(src, dst, s.n_pad, s.id[s.cur], s.id[s.cur ^ 1], s.n, st->nx, st->ny, st->nz, st->ntx,
                                                                        st->nty, st->ntiles, s.tile_start2[nw], s.tile_cursor);
    } else {
        bin3_scatter_kernel<T><<<nb, 256, shmem, h->stream>>>(src, dst, s.n_pad, s.id[s.cur], s.id[s.cur ^ 1], s.n, st->nx, st->ny, st->nz, st->ntx, st->nty,
                                                        st->ntiles, s.tile_start2[nw], s.tile_cursor);
    }
    HIP_TRY(h, hipGetLastError());
    if (!two_level) s.cur ^= 1;
    s.wl = nw;
    s.layout++;
    s.binned = true;
    s.ids_identity = false;
    s.census_fresh = s.rebin_pending = s.chunk_census_fresh = false; // tile_count now describes this binning, not a push
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

// ---- the library's own FFT passes (fes_fft.hpp)
// complex values per row of the half spectrum: nx / 2 + 1 for rocFFT's buffers; the library's own passes pad a row to whole
// column tiles (129 -> 144 floats, 136 doubles), so that a tile's piece of a row is one aligned 128-byte line (column
// passes 56 / 78 / 54 -> 45 / 72 / 39 us at 256^3, profiles/r03_fft_ablation.txt)
template <typename T>
size_t row_pitch(const State* st)
{
    const size_t nxh = st->nx / 2 + 1, c = fft_tile_columns<T>();
    return st->own_fft ? (nxh + c - 1) / c * c : nxh;
}

// complex values of the whole-grid transform buffer (sized before it is known whether these passes or rocFFT will use it)
template <typename T>
size_t hat_values(const State* st)
{
    const size_t c = fft_tile_columns<T>(), nxh = st->nx / 2 + 1;
    return (nxh + c - 1) / c * c * st->ny * st->nz;
}

// pairs of real rows per workgroup of the x passes (two rows ride on one complex transform): about 4096 points — 16 pairs
// of 256, 8 of 512 (512^3: x forward 525 -> 428 us, x inverse 353 -> 326 with 8 instead of 16; 4 pairs and, at 256^3, 8 or 4
// are slower: profiles/r03_fft_ablation.txt)
template <typename T>
int x_pairs_per_workgroup(int nx) { return std::max(2, std::min<int>(fft_tile_columns<T>(), 4096 / nx)); }

template <typename T>
int fft_x_forward(fpic_handle* h, const long long* fixed, const T* rho, double scale, size_t rows, T* hat)
{
    State* st = h->es;
    const int ppw = x_pairs_per_workgroup<T>(st->nx);
    fft_x_forward_kernel<T><<<blocks_for(rows, 2 * ppw), kFftThreads, fft_lds_bytes<T>(st->nx, ppw), h->stream>>>(fixed, rho, scale, rows, st->nx, fft_log2(st->nx), ppw, hat,
                                                                                                                 static_cast<const T*>(st->fft_tw[0]), static_cast<int>(row_pitch<T>(st)));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

template <typename T>
int fft_x_inverse(fpic_handle* h, const T* hat, size_t rows, T* phi)
{
    State* st = h->es;
    const int ppw = x_pairs_per_workgroup<T>(st->nx);
    fft_x_inverse_kernel<T><<<blocks_for(rows, 2 * ppw), kFftThreads, fft_lds_bytes<T>(st->nx, ppw), h->stream>>>(hat, rows, st->nx, fft_log2(st->nx), ppw, phi,
                                                                                                                 static_cast<const T*>(st->fft_tw[0]), static_cast<int>(row_pitch<T>(st)));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// columns of N points at `stride` complex elements, `outer` lines of them `outer_stride` apart (x fastest, nxh values);
// xbuf + nyl + nzl: the y passes of a slab-decomposed solve store into / load from the all-to-all's buffer (ColLayout)
template <typename T, int MODE>
int fft_columns(fpic_handle* h, T* hat, size_t outer_stride, size_t stride, int outer, int N, int y0 = 0, T* xbuf = nullptr, int nyl = 0, int nzl = 0)
{
    State* st = h->es;
    const T* twt = static_cast<const T*>(st->fft_tw[MODE == 2 ? 2 : 1]); // (the y passes and the z sweep: N is ny resp. nz)
    const int nxh = st->nx / 2 + 1;
    const ColLayout L{ outer_stride, stride, outer, nxh, static_cast<int>(row_pitch<T>(st)), xbuf ? nyl : 0, nzl };
    // (tiles of 16 complex floats / 8 doubles = one 128-byte line per row; at 256 points 4, 8 and 32 columns were measured
    // and lose.  Columns of 512 floats take half tiles: about 4096 points per workgroup again, as in the x passes — twice the
    // workgroups in flight: z sweep 674 -> 574 us, y inverse 385 -> 364 at 512^3, profiles/r03_fft_ablation.txt)
    constexpr int C = fft_tile_columns<T>();
    if (N >= 512 && sizeof(T) == 4) {
        constexpr int H = C / 2;
        const unsigned tiles = static_cast<unsigned>((nxh + H - 1) / H);
        fft_columns_kernel<T, MODE, H><<<static_cast<unsigned>(outer) * tiles, kFftThreads, fft_lds_bytes<T>(N, H), h->stream>>>(
            hat, xbuf, L, N, fft_log2(N), y0, st->k2[0], st->k2[1], st->k2[2], 1.0 / (kEps0 * static_cast<double>(st->nodes)), twt);
        HIP_TRY(h, hipGetLastError());
        return FPIC_OK;
    }
    const unsigned tiles = static_cast<unsigned>((nxh + C - 1) / C);
    fft_columns_kernel<T, MODE><<<static_cast<unsigned>(outer) * tiles, kFftThreads, fft_lds_bytes<T>(N, C), h->stream>>>(
        hat, xbuf, L, N, fft_log2(N), y0, st->k2[0], st->k2[1], st->k2[2], 1.0 / (kEps0 * static_cast<double>(st->nodes)), twt);
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// rho (T) of the whole grid / of a rank's own planes from the integer charge grid, when somebody reads it
template <typename T>
int refresh_rho(fpic_handle* h)
{
    State* st = h->es;
    if (st->rho_fresh) return FPIC_OK;
    const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
    const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv);
    size_t first = 0, count = st->nodes;
    if (const Domain* d = st->dom; d && d->world > 1) { first = lp(st, d->z0) * st->nx * st->ny; count = static_cast<size_t>(d->nzl) * st->nx * st->ny; }
    rho_real_kernel<T><<<blocks_for(count), 256, 0, h->stream>>>(st->rho_fixed + first, count, scale, static_cast<T*>(st->rho) + first);
    HIP_TRY(h, hipGetLastError());
    st->rho_fresh = true;
    return FPIC_OK;
}

// rho_fixed -> E4 (es3d_rho_real, es3d_poisson, es3d_gradient)
template <typename T>
int launch_solve(fpic_handle* h, bool convert = true)
{
    State* st = h->es;
    if (compact(st)) return fail(h, FPIC_ERR_STATE, "a rank with slab-only arrays solves with its group (the decomposed solve), not alone");
    timing_begin(h, KC_SOLVE);
    const double dv_ = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
    const double scale_ = h->spec.particle_charge * st->W / (4398046511104.0 * dv_); // q0 W / (2^42 dV)
    if (st->own_fft && (st->solver == FPIC_SOLVER_POISSON_FFT || st->solver == FPIC_SOLVER_YEE)) {
        // five sweeps: x forward (straight from the integer grid), y forward, the whole z direction with the k-space
        // factor, y inverse, x inverse
        const size_t nxh = row_pitch<T>(st); // (the rows' pitch)
        const size_t rows = static_cast<size_t>(st->ny) * st->nz, line = static_cast<size_t>(st->ny) * nxh;
        T* hat = static_cast<T*>(st->hat);
        int rc;
        if ((rc = fft_x_forward<T>(h, convert ? st->rho_fixed : nullptr, convert ? nullptr : static_cast<const T*>(st->rho), scale_, rows, hat)) ||
            (rc = fft_columns<T, 0>(h, hat, line, nxh, st->nz, st->ny)) ||
            (rc = fft_columns<T, 2>(h, hat, nxh, line, st->ny, st->nz)) ||
            (rc = fft_columns<T, 1>(h, hat, line, nxh, st->nz, st->ny)) ||
            (rc = fft_x_inverse<T>(h, hat, rows, static_cast<T*>(st->phi))))
            return rc;
        if (convert) st->rho_fresh = false;
        if (st->solver == FPIC_SOLVER_YEE) {
            em_edge_gradient_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(static_cast<const T*>(st->phi), st->nx, st->ny, st->nz,
                                                                                   static_cast<T>(1.0 / (st->lx / st->nx)), static_cast<T>(1.0 / (st->ly / st->ny)),
                                                                                   static_cast<T>(1.0 / (st->lz / st->nz)), static_cast<T*>(st->Ey), 0, st->nz, held_of(st));
            HIP_TRY(h, hipGetLastError());
        } else {
            gradient_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(
                static_cast<const T*>(st->phi), st->nx, st->ny, st->nz, static_cast<T>(1.0 / (2.0 * (st->lx / st->nx))),
                static_cast<T>(1.0 / (2.0 * (st->ly / st->ny))), static_cast<T>(1.0 / (2.0 * (st->lz / st->nz))), static_cast<T*>(st->E4));
            HIP_TRY(h, hipGetLastError());
        }
        timing_end(h);
        h->solve_launches++;
        return FPIC_OK;
    }
    if (convert) { // (a decomposed run has converted its own planes and gathered the others)
        const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
        const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv); // q0 W / (2^42 dV)
        rho_real_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(st->rho_fixed, st->nodes, scale, static_cast<T*>(st->rho));
        HIP_TRY(h, hipGetLastError());
        st->rho_fresh = true;
    }
    if (st->solver == FPIC_SOLVER_POISSON_FFT || st->solver == FPIC_SOLVER_YEE) {
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
        if (st->solver == FPIC_SOLVER_YEE) // the field on the lattice's edges: Gauss's law holds exactly there
            em_edge_gradient_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(static_cast<const T*>(st->phi), st->nx, st->ny, st->nz,
                                                                                   static_cast<T>(1.0 / (st->lx / st->nx)), static_cast<T>(1.0 / (st->ly / st->ny)),
                                                                                   static_cast<T>(1.0 / (st->lz / st->nz)), static_cast<T*>(st->Ey), 0, st->nz, held_of(st));
        else
            gradient_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(
                static_cast<const T*>(st->phi), st->nx, st->ny, st->nz, static_cast<T>(1.0 / (2.0 * (st->lx / st->nx))),
                static_cast<T>(1.0 / (2.0 * (st->ly / st->ny))), static_cast<T>(1.0 / (2.0 * (st->lz / st->nz))), static_cast<T*>(st->E4));
        HIP_TRY(h, hipGetLastError());
    }
    timing_end(h);
    h->solve_launches++;
    return FPIC_OK;
}

// Interior tile layers of a rank's slab: at least one whole layer of tiles (2^ltz planes) away from either face, so that
// nothing a particle of an interior tile deposits can reach a plane that is exchanged (a particle drifts at most G planes
// between two migrations, and G + 1 (electrostatic) / G + 2 (full EM) <= 2^ltz is asked for).  Empty range: no split.
bool interior_layers(const State* st, uint32_t& lo, uint32_t& hi)
{
    const Domain* d = st->dom;
    lo = hi = 0;
    if (!d || d->world < 2 || !d->overlap) return false;
    const int tz = 1 << st->ltz;
    const int reach = st->solver == FPIC_SOLVER_YEE ? d->G + 2 : d->G + 1;
    if (reach > tz) return false;
    const int first = (d->z0 + tz - 1) / tz + 1, last = (d->z0 + d->nzl) / tz - 1; // [first, last)
    if (first >= last) return false;
    lo = static_cast<uint32_t>(first); hi = static_cast<uint32_t>(last);
    return true;
}

// every species binned (the work list is in tile order) and a non-empty interior: the push may go in two parts
bool can_split(const State* st)
{
    uint32_t lo, hi;
    if (!interior_layers(st, lo, hi)) return false;
    for (const Species& s : st->sp)
        if (!s.binned) return false;
    return true;
}

// part 0: memsets and every species in one go; part 1: memsets and the first part of every species; part 2: the rest
template <typename T, bool DEPOSIT_ONLY>
int deposit_cycle(fpic_handle* h, int part = 0)
{
    State* st = h->es;
    if (part != 2) {
        timing_begin(h, DEPOSIT_ONLY ? KC_DEPOSIT : KC_PUSH);
        if (st->dom && st->dom->world > 1) {
            // a rank of a decomposition deposits on its own planes and the ghost planes only (whatever a particle that has
            // outrun them adds elsewhere is never read): planes [z0 - G, z0 + nzl + G], periodic
            const Domain& d = *st->dom;
            const size_t plane = static_cast<size_t>(st->nx) * st->ny; // nodes
            if (int rc = zero_planes(h, st->rho_fixed, plane * sizeof(long long), d.z0 - d.G, d.nzl + 2 * d.G + 1)) return rc;
        } else {
            HIP_TRY(h, hipMemsetAsync(st->rho_fixed, 0, st->nodes * sizeof(long long), h->stream));
        }
        HIP_TRY(h, hipMemsetAsync(st->spilled, 0, sizeof(unsigned long long), h->stream));
    }
    const int rc = launch_push_all<T, DEPOSIT_ONLY>(h, part);
    if (part != 1) timing_end(h);
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
    // (a re-binning launch still works in the OLD tiles' windows: its count of deposits outside them is the reason it was
    // asked for, not a reading of the new order — taken as one, it asked for a second re-binning two sub-steps later)
    if (!rebin) {
        HIP_TRY(h, hipMemcpyAsync(st->spilled_host + slot, st->spilled, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipEventRecord(st->spill_event[slot], h->stream));
        st->spill_pending[slot] = true;
    }
    st->substeps_since_bin++;
    h->step_launches++;
    h->particle_updates += total_particles(st);
    return launch_solve<T>(h);
}

// ---- full EM (solver = YEE): em_nodes, em_push + em_current, B half, E, B half (oracle: em_substep)
template <typename T>
int em_nodes(fpic_handle* h, int k0 = 0, int nk = -1)
{
    State* st = h->es;
    if (nk < 0 && compact(st)) { k0 = st->zs0 + 1; nk = st->nzs - 1; } // (a node reads the edges and faces of the plane below as well)
    if (nk < 0 || nk > st->nz) nk = st->nz;
    k0 = (k0 % st->nz + st->nz) % st->nz;
    em_nodes_kernel<T><<<blocks_for(static_cast<size_t>(st->nx) * st->ny * nk), 256, 0, h->stream>>>(static_cast<const T*>(st->Ey), static_cast<const T*>(st->By), st->nx,
                                                                                                  st->ny, st->nz, static_cast<T*>(st->E4), static_cast<T*>(st->B4n), k0, nk,
                                                                                                  held_of(st));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

bool can_split(const State* st);
bool interior_layers(const State* st, uint32_t& lo, uint32_t& hi);

// the currents of one sub-step: gather + Boris + move + integer current deposit of every species (Jfix zeroed by the caller)
template <typename T>
int em_push_all(fpic_handle* h, int part = 0)
{
    State* st = h->es;
    const double dt = h->spec.dt;
    for (Species& s : st->sp) {
        if (!s.n) continue;
        EmPushArgs<T> a{};
        a.slab = static_cast<T*>(s.slab[s.cur]); a.stride = s.n_pad; a.n = s.n;
        a.E4n = static_cast<const T*>(st->E4); a.B4n = static_cast<const T*>(st->B4n);
        a.Jfix = reinterpret_cast<unsigned long long*>(st->Jfix);
        a.nx = st->nx; a.ny = st->ny; a.nz = st->nz;
        a.held = held_of(st);
        const double hh = s.charge * dt / (2 * s.mass), step = dt * kSpeedOfLight;
        a.h = static_cast<T>(hh);
        a.hc = static_cast<T>(hh) / static_cast<T>(kSpeedOfLight); // in T, as the oracle forms it
        a.dx = static_cast<T>(step / st->lx); a.dy = static_cast<T>(step / st->ly); a.dz = static_cast<T>(step / st->lz);
        a.Z = s.Z;
        if (s.binned) {
            EmTileArgs<T> t{};
            t.p = a;
            t.ntx = st->ntx; t.nty = st->nty; t.ntz = st->ntz;
            t.work = s.work2[s.wl]; t.nwork = s.nwork2[s.wl];
            t.part = part; t.tiles_per_layer = static_cast<uint32_t>(st->ntx) * st->nty;
            interior_layers(st, t.layer_lo, t.layer_hi);
            t.spilled = st->spilled;
            t.tile_start = s.tile_start2[s.wl];
            if (part != 2) { // (the second part of a split push follows the first on this stream with the same grid)
                if (!s.em_args)
                    if (int rc = dev_alloc(h, &s.em_args, sizeof(EmPushArgs<double>), &h->bytes_grid)) return rc;
                store_args_kernel<EmPushArgs<T>><<<1, 1, 0, h->stream>>>(a, static_cast<EmPushArgs<T>*>(s.em_args));
            }
            t.resident = static_cast<const EmPushArgs<T>*>(s.em_args);
            // (FES_EM_PIPE: a persistent workgroup per CU walks the list with two windows; otherwise one workgroup per item)
            const unsigned grid = FES_EM_PIPE ? static_cast<unsigned>(std::min<size_t>(s.work_cap, static_cast<size_t>(h->cus))) : static_cast<unsigned>(s.work_cap);
            em_push_tiles_kernel<T><<<grid, em_threads<T>(), em_lds_bytes<T>(), h->stream>>>(t);
        } else if (part != 2) { // (an unbinned species is pushed whole with the first part)
            em_push_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(a);
        }
        HIP_TRY(h, hipGetLastError());
    }
    return FPIC_OK;
}

// the lattice update, piecewise: planes [k0, k0 + nk) (periodic) of B (half a step) or E (a step)
template <typename T>
struct EmCoef {
    T cb[3], ce[3], je;
    double js[3];
    EmCoef(const fpic_handle* h)
    {
        const State* st = h->es;
        const double dt = h->spec.dt;
        const double d[3] = { st->lx / st->nx, st->ly / st->ny, st->lz / st->nz };
        const double c2 = kSpeedOfLight * kSpeedOfLight;
        const double base = h->spec.particle_charge * st->W / (96.0 * 4398046511104.0 * dt);
        for (int a = 0; a < 3; ++a) { cb[a] = static_cast<T>(dt / (2 * d[a])); ce[a] = static_cast<T>(c2 * dt / d[a]); }
        je = static_cast<T>(dt / kEps0);
        js[0] = base / (d[1] * d[2]); js[1] = base / (d[0] * d[2]); js[2] = base / (d[0] * d[1]);
    }
};

// the two half-time arrays of the chained lattice step: both or none (a launch must never see one of them null)
template <typename T>
int alloc_half_time(fpic_handle* h, size_t nodes)
{
    State* st = h->es;
    for (int k = 0; k < 2; ++k) {
        if (st->Bh[k]) continue;
        if (int rc = dev_alloc(h, &st->Bh[k], nodes * 4 * sizeof(T), &h->bytes_grid)) {
            for (int j = 0; j < 2; ++j)
                if (st->Bh[j]) { (void)hipFree(st->Bh[j]); st->Bh[j] = nullptr; h->bytes_grid -= nodes * 4 * sizeof(T); }
            return rc;
        }
    }
    return FPIC_OK;
}

template <typename T>
int em_half_b(fpic_handle* h, const EmCoef<T>& c, int k0, int nk, const void* from = nullptr, void* to = nullptr)
{
    State* st = h->es;
    k0 = (k0 % st->nz + st->nz) % st->nz;
    em_update_b_kernel<T><<<blocks_for(static_cast<size_t>(st->nx) * st->ny * nk), 256, 0, h->stream>>>(static_cast<T*>(to ? to : st->By), static_cast<const T*>(st->Ey), st->nx, st->ny,
                                                                                                     st->nz, c.cb[0], c.cb[1], c.cb[2], k0, nk, held_of(st),
                                                                                                     static_cast<const T*>(from));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// the chained lattice step (em_chain_tiled_kernel) on the node planes k0 .. k0 + nk - 1: Bh[bh_cur], Ey -> E4, B4n, Bh[bh_cur ^ 1]
template <typename T>
int em_chain_launch(fpic_handle* h, const EmCoef<T>& co, int k0, int nk, bool below_too)
{
    State* st = h->es;
    k0 = (k0 % st->nz + st->nz) % st->nz;
    const unsigned tiles = static_cast<unsigned>(((st->nx + kCX - 1) / kCX) * ((st->ny + kCY - 1) / kCY) * ((nk + kCZ - 1) / kCZ));
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(em_chain_tiled_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(em_chain_lds_bytes<T>())));
    em_chain_tiled_kernel<T><<<tiles, kChainThreads, em_chain_lds_bytes<T>(), h->stream>>>(static_cast<const T*>(st->Bh[st->bh_cur]), static_cast<const T*>(st->Ey), st->nx, st->ny,
                                                                                          st->nz, co.cb[0], co.cb[1], co.cb[2], static_cast<T*>(st->E4), static_cast<T*>(st->B4n),
                                                                                          static_cast<T*>(st->Bh[st->bh_cur ^ 1]), k0, nk, held_of(st), below_too ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

// By <- the B of the integer time the chained step has reached (the half step it left open); whoever reads or replaces By calls it
template <typename T>
int em_close(fpic_handle* h)
{
    State* st = h->es;
    if (!st->em_open) return FPIC_OK;
    const EmCoef<T> co(h);
    if (int rc = em_half_b<T>(h, co, 0, st->nz, st->Bh[st->bh_cur], st->By)) return rc;
    st->em_open = false;
    return FPIC_OK;
}
int em_close_any(fpic_handle* h)
{
    if (!h->es || !h->es->em_open) return FPIC_OK;
    return h->prec == FPIC_F32 ? em_close<float>(h) : em_close<double>(h);
}

template <typename T>
int em_full_e(fpic_handle* h, const EmCoef<T>& c, int k0, int nk, const void* b = nullptr)
{
    State* st = h->es;
    k0 = (k0 % st->nz + st->nz) % st->nz;
    em_update_e_kernel<T><<<blocks_for(static_cast<size_t>(st->nx) * st->ny * nk), 256, 0, h->stream>>>(static_cast<T*>(st->Ey), static_cast<const T*>(b ? b : st->By), st->Jfix, st->nx,
                                                                                                     st->ny, st->nz, c.ce[0], c.ce[1], c.ce[2], c.je, c.js[0], c.js[1],
                                                                                                     c.js[2], k0, nk, held_of(st));
    HIP_TRY(h, hipGetLastError());
    return FPIC_OK;
}

template <typename T>
int em_precalc(fpic_handle* h)
{
    State* st = h->es;
    // A large fresh population is binned first, as in precalc() of the electrostatic cycle: the first sub-step would bin it
    // anyway, and although the charge grid of this mode is deposited in the flat form (a diagnostic, and the start field),
    // particles in tile order add to neighbouring nodes — the atomics of a wave meet in a few cache lines instead of 512.
    bool bin_first = false;
    for (const Species& sp : st->sp) bin_first |= !sp.binned && sp.n >= h->two_level_min;
    if (bin_first)
        if (int rc = bin_all<T>(h, true)) return rc;
    if (int rc = deposit_cycle<T, true>(h)) return rc;
    h->deposit_launches++;
    if (int rc = launch_solve<T>(h)) return rc; // rho -> phi -> E on the edges
    st->em_open = false;                        // (both lattice fields are set afresh)
    fill4_kernel<T><<<blocks_for(held_nodes(st)), 256, 0, h->stream>>>(static_cast<T*>(st->By), held_nodes(st), static_cast<T>(st->B0[0]), static_cast<T>(st->B0[1]),
                                                                     static_cast<T>(st->B0[2]));
    HIP_TRY(h, hipGetLastError());
    return em_nodes<T>(h);
}

template <typename T>
int em_substep(fpic_handle* h)
{
    State* st = h->es;
    bool unbinned = false;
    for (const Species& s : st->sp) unbinned |= !s.binned;
    // re-bin (three-pass form) when currents start to miss the LDS window (lagged read-back) or after 64 sub-steps: an
    // EM step moves a thermal particle by a small fraction of a cell (c dt < dx / sqrt 3)
    bool rebin = unbinned || st->substeps_since_bin >= 64;
    if (!rebin) {
        const int slot = static_cast<int>(st->spill_seq & 1);
        if (st->spill_pending[slot]) {
            HIP_TRY(h, hipEventSynchronize(st->spill_event[slot]));
            st->last_spill = st->spilled_host[slot];
            st->spill_pending[slot] = false;
        }
        rebin = st->last_spill * 512 > total_particles(st);
    }
    if (rebin)
        if (int rc = bin_all<T>(h, true)) return rc;
    HIP_TRY(h, hipMemsetAsync(st->spilled, 0, sizeof(unsigned long long), h->stream));
    // The lattice in two sweeps per sub-step instead of four (round 4): the sub-step's second B half step, the next sub-step's
    // node centring and its first B half step are ONE kernel (em_chain_tiled_kernel) between two half-time arrays; B at the
    // integer time is formed when somebody asks for it (em_close).  Bit-identical to the four sweeps; 13.0 against 13.7 ms per
    // sub-step at 256^3 / 5e8 fp64 (profiles/r04_em_chain_ablation.txt).  FPIC_EM_CHAIN=0 keeps the four sweeps, =flat the
    // form without LDS (one thread per node, 48 cached loads each: bound by the L1, no faster than four sweeps).
    const char* chain_env = std::getenv("FPIC_EM_CHAIN");
    if (!chain_env) chain_env = "1";
    const bool chain = std::strcmp(chain_env, "0") != 0 && !st->dom;
    const EmCoef<T> co(h);
    if (!chain && st->em_open)      // (the switch was turned off between two sub-steps)
        if (int rc = em_close<T>(h)) return rc;
    if (chain && (!st->Bh[0] || !st->Bh[1]))
        if (int rc = alloc_half_time<T>(h, st->nodes)) return rc;
    if (chain && st->em_open) {
        timing_begin(h, KC_SOLVE);
        if (std::strcmp(chain_env, "flat") == 0) {
            em_chain_kernel<T><<<blocks_for(st->nodes), 256, 0, h->stream>>>(static_cast<const T*>(st->Bh[st->bh_cur]), static_cast<const T*>(st->Ey), st->nx, st->ny, st->nz,
                                                                           co.cb[0], co.cb[1], co.cb[2], static_cast<T*>(st->E4), static_cast<T*>(st->B4n),
                                                                           static_cast<T*>(st->Bh[st->bh_cur ^ 1]));
        } else {
            if (int rc = em_chain_launch<T>(h, co, 0, st->nz, false)) return rc;
        }
        HIP_TRY(h, hipGetLastError());
        timing_end(h);
        st->bh_cur ^= 1;
    } else if (int rc = em_nodes<T>(h)) {
        return rc;
    }
    timing_begin(h, KC_PUSH);
    HIP_TRY(h, hipMemsetAsync(st->Jfix, 0, st->nodes * 3 * sizeof(long long), h->stream));
    if (int rc = em_push_all<T>(h)) return rc;
    timing_end(h);
    {
        const int slot = static_cast<int>(st->spill_seq++ & 1);
        HIP_TRY(h, hipMemcpyAsync(st->spilled_host + slot, st->spilled, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipEventRecord(st->spill_event[slot], h->stream));
        st->spill_pending[slot] = true;
    }
    timing_begin(h, KC_SOLVE);
    if (chain) {
        if (!st->em_open) { // from B at the integer time: its first half step goes to the half-time array, By goes stale
            st->bh_cur = 0;
            if (int rc = em_half_b<T>(h, co, 0, st->nz, st->By, st->Bh[0])) return rc;
            st->em_open = true;
        }
        if (int rc = em_full_e<T>(h, co, 0, st->nz, st->Bh[st->bh_cur])) return rc;
    } else {
        if (int rc = em_half_b<T>(h, co, 0, st->nz)) return rc;
        if (int rc = em_full_e<T>(h, co, 0, st->nz)) return rc;
        if (int rc = em_half_b<T>(h, co, 0, st->nz)) return rc;
    }
    timing_end(h);
    st->substeps_since_bin++;
    h->step_launches++;
    h->solve_launches++;
    h->particle_updates += total_particles(st);
    return FPIC_OK;
}

// `host` (host or device memory) holds the caller's particles [first, first + count)
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
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyDefault, h->stream); // host or device memory
        if (e == hipSuccess) {
            if (s.ids_identity)
                set_pos3_kernel<T, In><<<blocks_for(m), 256, 0, h->stream>>>(stage, first + b, m, 1 / st->lx, 1 / st->ly, 1 / st->lz, a, a + s.n_pad, a + 2 * s.n_pad,
                                                                           nullptr, first + b + m, first + b);
            else
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
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyDefault, h->stream); // host or device memory
        if (e == hipSuccess) {
            // velocities stay in units of c, unscaled
            if (s.ids_identity)
                set_vec3_kernel<T, In><<<blocks_for(m), 256, 0, h->stream>>>(stage, first + b, m, 1.0, 1.0, a + 3 * s.n_pad, a + 4 * s.n_pad, a + 5 * s.n_pad,
                                                                           nullptr, nullptr, first + b + m, first + b);
            else
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
int download_vec3(fpic_handle* h, const Species& s, Out* host, int first, size_t from = 0, size_t count = ~size_t(0), size_t stride = 1)
{
    // `count` of the caller's particles from, from + stride, ... (everything by default), in pieces of `chunk` output slots
    if (count == ~size_t(0)) count = s.n;
    if (!count) return FPIC_OK;
    const size_t chunk = 8u << 20;
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(Out)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        get_vec3_kernel<T, Out><<<blocks_for(s.n), 256, 0, h->stream>>>(a + first * s.n_pad, a + (first + 1) * s.n_pad, a + (first + 2) * s.n_pad,
                                                                      s.id[s.cur], s.n, b, m, stage, from, stride);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host + 3 * b, stage, m * 3 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T>
int download_cells(fpic_handle* h, const Species& s, int32_t* cells, size_t from = 0, size_t count = ~size_t(0), size_t stride = 1)
{
    State* st = h->es;
    if (count == ~size_t(0)) count = s.n;
    if (!count) return FPIC_OK;
    const size_t chunk = 16u << 20;
    int32_t* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * sizeof(int32_t)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        cells3_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(a, a + s.n_pad, a + 2 * s.n_pad, s.id[s.cur], s.n, b, m, st->nx, st->ny, st->nz, stage, from, stride);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(cells + b, stage, m * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "cell read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_field(fpic_handle* h, const In* host, void* target)
{
    State* st = h->es;
    In* stage = nullptr;
    const size_t bytes = st->nodes * 3 * sizeof(In);
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), bytes));
    hipError_t e = hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        pack_field3_kernel<T, In><<<blocks_for(st->nodes), 256, 0, h->stream>>>(stage, st->nx, st->ny, st->nz, static_cast<T*>(target), held_of(st));
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
    int rc;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&st->rho_fixed), st->nodes * sizeof(long long), acc)) ||
        (rc = dev_alloc(h, &st->rho, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->phi, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->E4, st->nodes * 4 * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->hat, hat_values<T>(st) * 2 * sizeof(T), acc)) ||
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
        (e = set_lds(bin3_scatter_kernel<T>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_count_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_scatter_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(em_push_tiles_kernel<T>, em_lds_bytes<T>())) != hipSuccess)
        return fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));

    if (st->solver == FPIC_SOLVER_YEE) {
        if ((rc = dev_alloc(h, &st->Ey, st->nodes * 4 * sizeof(T), acc)) || (rc = dev_alloc(h, &st->By, st->nodes * 4 * sizeof(T), acc)) ||
            (rc = dev_alloc(h, &st->B4n, st->nodes * 4 * sizeof(T), acc)) ||
            (rc = dev_alloc(h, reinterpret_cast<void**>(&st->Jfix), st->nodes * 3 * sizeof(long long), acc)))
            return rc;
    }
    // power-of-two grids (8 .. 512 nodes per axis): the library's own FFT passes; FPIC_POISSON_FFT=rocfft keeps rocFFT (a
    // development switch: the two agree within the solve's tolerance, tests/test_gpu_es3d.py)
    {
        const char* force = std::getenv("FPIC_POISSON_FFT");
        st->own_fft = fft_supported(st->nx) && fft_supported(st->ny) && fft_supported(st->nz) && !(force && std::strcmp(force, "rocfft") == 0);
        if (st->own_fft) {
            const int dims3[3] = { st->nx, st->ny, st->nz };
            for (int a = 0; a < 3; ++a) {
                if ((rc = dev_alloc(h, &st->fft_tw[a], static_cast<size_t>(dims3[a]) * 2 * sizeof(T), acc))) return rc;
                fft_twiddle_table_kernel<T><<<blocks_for(dims3[a]), 256, 0, h->stream>>>(static_cast<T*>(st->fft_tw[a]), dims3[a]);
            }
            HIP_TRY(h, hipGetLastError());
        }
        const size_t most = fft_lds_bytes<T>(1 << kFftMaxLog, fft_tile_columns<T>());
        // (fft_columns() launches the half-width instantiation for 512-point float columns: every form it can launch gets
        // its limit, whatever a later retune of the tile widths makes of their sizes)
        constexpr int CH = fft_tile_columns<T>() / 2;
        if ((e = set_lds(fft_x_forward_kernel<T>, most)) != hipSuccess || (e = set_lds(fft_x_inverse_kernel<T>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 0>, most)) != hipSuccess || (e = set_lds(fft_columns_kernel<T, 1>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 2>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 0, CH>, most)) != hipSuccess || (e = set_lds(fft_columns_kernel<T, 1, CH>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 2, CH>, most)) != hipSuccess)
            return fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    if ((st->solver == FPIC_SOLVER_POISSON_FFT || st->solver == FPIC_SOLVER_YEE) && !st->own_fft) { // (YEE: the initial field is the Poisson field)
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

// ---- checkpoint of an undecomposed box: header, per species the raw particle state in the caller's order, the fields
namespace {

// format version of both checkpoint files: 2 since the header's fpic_spec is the one of ABI 2 (a file written by an
// older library is refused by its version, not as "truncated")
constexpr uint32_t kCheckpointVersion = 2;

struct BoxCheckpointHeader {
    char magic[8];        // "FPICBOX1"
    uint32_t version;     // kCheckpointVersion
    uint32_t precision, solver, nspecies;
    int32_t nx, ny, nz;
    uint32_t fields_ready;
    double B0[3];
    fpic_spec spec;
};
struct BoxCheckpointSpecies {
    uint64_t n;
    double mass, charge;
};

struct BoxFile {
    FILE* f;
    ~BoxFile() { if (f) std::fclose(f); }
};

// the device arrays a checkpoint carries besides the particles
std::vector<std::pair<void*, size_t>> checkpoint_fields(const fpic_handle* h)
{
    const State* st = h->es;
    const size_t t = h->prec == FPIC_F32 ? 4 : 8;
    std::vector<std::pair<void*, size_t>> out;
    out.push_back({ st->E4, st->nodes * 4 * t });
    if (st->solver == FPIC_SOLVER_YEE) {
        out.push_back({ st->Ey, st->nodes * 4 * t });
        out.push_back({ st->By, st->nodes * 4 * t });
        out.push_back({ st->B4n, st->nodes * 4 * t });
    }
    return out;
}

constexpr size_t kCkptChunk = size_t(4) << 20; // particles per staging round (96 / 192 MB)

template <typename T>
int checkpoint_particles(fpic_handle* h, FILE* f, bool save)
{
    State* st = h->es;
    T* stage = nullptr;
    size_t most = 0;
    for (const Species& s : st->sp) most = std::max(most, std::min(kCkptChunk, s.n));
    if (!most) return FPIC_OK;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), most * 6 * sizeof(T)));
    std::vector<T> host(most * 6);
    int rc = FPIC_OK;
    for (Species& s : st->sp) {
        for (size_t first = 0; first < s.n && rc == FPIC_OK; first += kCkptChunk) {
            const size_t m = std::min(kCkptChunk, s.n - first);
            hipError_t e = hipSuccess;
            if (save) {
                ckpt_gather_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(static_cast<const T*>(s.slab[s.cur]), s.n_pad, s.id[s.cur], s.n, first, m, stage);
                if ((e = hipGetLastError()) == hipSuccess) e = hipMemcpyAsync(host.data(), stage, m * 6 * sizeof(T), hipMemcpyDeviceToHost, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e == hipSuccess && std::fwrite(host.data(), sizeof(T), m * 6, f) != m * 6) rc = fail(h, FPIC_ERR_STATE, "checkpoint write failed");
            } else {
                if (std::fread(host.data(), sizeof(T), m * 6, f) != m * 6) { rc = fail(h, FPIC_ERR_STATE, "checkpoint read failed"); break; }
                e = hipMemcpyAsync(stage, host.data(), m * 6 * sizeof(T), hipMemcpyHostToDevice, h->stream);
                if (e == hipSuccess) {
                    ckpt_scatter_kernel<T><<<blocks_for(m), 256, 0, h->stream>>>(static_cast<T*>(s.slab[s.cur]), s.n_pad, s.id[s.cur], first, m, stage);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            }
            if (e != hipSuccess) rc = fail(h, FPIC_ERR_HIP, "checkpoint transfer failed: %s", hipGetErrorString(e));
        }
        if (rc) break;
    }
    (void)hipFree(stage);
    return rc;
}

int checkpoint_arrays(fpic_handle* h, FILE* f, bool save)
{
    std::vector<unsigned char> host(size_t(64) << 20);
    for (const auto& a : checkpoint_fields(h)) {
        for (size_t off = 0; off < a.second; off += host.size()) {
            const size_t m = std::min(host.size(), a.second - off);
            if (save) {
                HIP_TRY(h, hipMemcpyAsync(host.data(), static_cast<const char*>(a.first) + off, m, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                if (std::fwrite(host.data(), 1, m, f) != m) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
            } else {
                if (std::fread(host.data(), 1, m, f) != m) return fail(h, FPIC_ERR_STATE, "checkpoint read failed");
                HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(a.first) + off, host.data(), m, hipMemcpyHostToDevice, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
            }
        }
    }
    return FPIC_OK;
}

} // namespace

// ---- checkpoint of ONE RANK of a decomposition (every rank writes its own file): header, per species the particles the
// rank holds now — global indices and raw state in slot order — and the state that cannot be recomputed: the lattice
// fields of the rank's own planes (full EM) or the given field (solver 'none').  An electrostatic run with the Poisson
// solve stores no field: after the load precalc() recomputes it from the particles, bit for bit.
namespace {

struct RankCheckpointHeader {
    char magic[8];        // "FPICRNK1"
    uint32_t version, precision, solver, nspecies;
    int32_t nx, ny, nz, rank, world, ghost_planes;
    uint32_t fields_ready, reserved;
    double B0[3];
    fpic_spec spec;
};

int rank_io(fpic_handle* h, FILE* f, void* dev, size_t bytes, bool save)
{
    std::vector<unsigned char> host(std::min<size_t>(bytes, size_t(64) << 20));
    for (size_t off = 0; off < bytes; off += host.size()) {
        const size_t m = std::min(host.size(), bytes - off);
        if (save) {
            HIP_TRY(h, hipMemcpyAsync(host.data(), static_cast<const char*>(dev) + off, m, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (std::fwrite(host.data(), 1, m, f) != m) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
        } else {
            if (std::fread(host.data(), 1, m, f) != m) return fail(h, FPIC_ERR_STATE, "checkpoint read failed");
            HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(dev) + off, host.data(), m, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
    }
    return FPIC_OK;
}

// the field arrays a rank's file carries: (device pointer, first byte, bytes)
std::vector<std::pair<char*, size_t>> rank_fields(const fpic_handle* h)
{
    const State* st = h->es;
    const Domain& d = *st->dom;
    const size_t t = h->prec == FPIC_F32 ? 4 : 8, plane = static_cast<size_t>(st->nx) * st->ny;
    std::vector<std::pair<char*, size_t>> out;
    if (st->solver == FPIC_SOLVER_YEE) {
        out.push_back({ static_cast<char*>(st->Ey) + 4 * t * plane * lp(st, d.z0), 4 * t * plane * d.nzl });
        out.push_back({ static_cast<char*>(st->By) + 4 * t * plane * lp(st, d.z0), 4 * t * plane * d.nzl });
    } else if (st->solver == FPIC_SOLVER_NONE) {
        out.push_back({ static_cast<char*>(st->E4), 4 * t * st->nodes });
    }
    return out;
}

int save_rank_checkpoint(fpic_handle* h, const char* path)
{
    State* st = h->es;
    const Domain& d = *st->dom;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    BoxFile bf{ std::fopen(path, "wb") };
    if (!bf.f) return fail(h, FPIC_ERR_STATE, "cannot open %s for writing", path);
    RankCheckpointHeader hd{};
    std::memcpy(hd.magic, "FPICRNK1", 8);
    hd.version = kCheckpointVersion; hd.precision = static_cast<uint32_t>(h->prec); hd.solver = static_cast<uint32_t>(st->solver);
    hd.nspecies = static_cast<uint32_t>(st->sp.size());
    hd.nx = st->nx; hd.ny = st->ny; hd.nz = st->nz; hd.rank = d.rank; hd.world = d.world; hd.ghost_planes = d.G;
    hd.fields_ready = st->fields_ready ? 1 : 0;
    for (int a = 0; a < 3; ++a) hd.B0[a] = st->B0[a];
    hd.spec = h->spec;
    if (std::fwrite(&hd, sizeof hd, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    const size_t t = h->prec == FPIC_F32 ? 4 : 8;
    for (const Species& s : st->sp) {
        const BoxCheckpointSpecies bs{ s.n, s.mass, s.charge };
        if (std::fwrite(&bs, sizeof bs, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    }
    for (const Species& s : st->sp) {
        if (!s.n) continue;
        if (int rc = rank_io(h, bf.f, s.id[s.cur], s.n * sizeof(uint32_t), true)) return rc;
        for (int f = 0; f < 6; ++f)
            if (int rc = rank_io(h, bf.f, static_cast<char*>(s.slab[s.cur]) + f * s.n_pad * t, s.n * t, true)) return rc;
    }
    for (const auto& a : rank_fields(h))
        if (int rc = rank_io(h, bf.f, a.first, a.second, true)) return rc;
    return FPIC_OK;
}

int load_rank_checkpoint(fpic_handle* h, const char* path)
{
    State* st = h->es;
    Domain& d = *st->dom;
    BoxFile bf{ std::fopen(path, "rb") };
    if (!bf.f) return fail(h, FPIC_ERR_STATE, "cannot open %s", path);
    RankCheckpointHeader hd{};
    if (std::fread(&hd, 12, 1, bf.f) != 1 || std::memcmp(hd.magic, "FPICRNK1", 8) != 0)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is not a checkpoint of a rank of a decomposed box", path);
    if (hd.version != kCheckpointVersion && hd.version != 1) // (version 1: the same layout, written before the number was raised)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is a rank checkpoint of format version %u; this library reads version %u (the header embeds fpic_spec of ABI %d)", path, hd.version,
                    kCheckpointVersion, FPIC_ABI_VERSION);
    if (std::fread(reinterpret_cast<char*>(&hd) + 12, sizeof hd - 12, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
    if (static_cast<int>(hd.precision) != h->prec || static_cast<int>(hd.solver) != st->solver || hd.nspecies != st->sp.size() || hd.nx != st->nx || hd.ny != st->ny ||
        hd.nz != st->nz || hd.rank != d.rank || hd.world != d.world)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint of rank %d of %d, %u species on %d x %d x %d, precision %u, solver %u; this is rank %d of %d, %zu species on %d x %d x %d, precision %d, solver %d",
                    hd.rank, hd.world, hd.nspecies, hd.nx, hd.ny, hd.nz, hd.precision, hd.solver, d.rank, d.world, st->sp.size(), st->nx, st->ny, st->nz, h->prec, st->solver);
    if (hd.spec.radius != h->spec.radius || hd.spec.length_y != h->spec.length_y || hd.spec.height != h->spec.height || hd.spec.dt != h->spec.dt ||
        hd.spec.macro_weight != h->spec.macro_weight)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint was written with different lengths / dt / macro_weight");
    const size_t t = h->prec == FPIC_F32 ? 4 : 8;
    std::vector<BoxCheckpointSpecies> bs(st->sp.size());
    unsigned long long want = sizeof hd + bs.size() * sizeof(BoxCheckpointSpecies);
    for (size_t k = 0; k < bs.size(); ++k) {
        if (std::fread(&bs[k], sizeof bs[k], 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
        if (bs[k].mass != st->sp[k].mass || bs[k].charge != st->sp[k].charge)
            return fail(h, FPIC_ERR_INVALID_ARG, ".species <- species %zu of the checkpoint (mass %g, charge %g) is not the pusher's", k, bs[k].mass, bs[k].charge);
        if (bs[k].n > st->sp[k].cap)
            return fail(h, FPIC_ERR_INVALID_ARG, ".species <- the checkpoint holds %llu particles of species %zu, the rank's capacity is %zu", static_cast<unsigned long long>(bs[k].n), k, st->sp[k].cap);
        want += bs[k].n * (sizeof(uint32_t) + 6ull * t);
    }
    for (const auto& a : rank_fields(h)) want += a.second;
    const long at = std::ftell(bf.f);
    if (at < 0 || std::fseek(bf.f, 0, SEEK_END) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    const long long have = std::ftell(bf.f);
    if (have < 0 || static_cast<unsigned long long>(have) < want) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated: %lld bytes, %llu expected", have, want);
    if (std::fseek(bf.f, at, SEEK_SET) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t k = 0; k < bs.size(); ++k) {
        Species& s = st->sp[k];
        s.n = static_cast<size_t>(bs[k].n);
        s.binned = s.census_fresh = s.rebin_pending = false; // slot order of the file: the first sub-step bins
        s.ids_identity = false;
        s.tail_first = s.tail_count = s.n_after = 0;
        if (!s.n) continue;
        if (int rc = rank_io(h, bf.f, s.id[s.cur], s.n * sizeof(uint32_t), false)) return rc;
        for (int f = 0; f < 6; ++f)
            if (int rc = rank_io(h, bf.f, static_cast<char*>(s.slab[s.cur]) + f * s.n_pad * t, s.n * t, false)) return rc;
    }
    for (const auto& a : rank_fields(h))
        if (int rc = rank_io(h, bf.f, a.first, a.second, false)) return rc;
    for (int a = 0; a < 3; ++a) st->B0[a] = hd.B0[a];
    st->em_open = false;    // (the file's B is B of the integer time)
    st->spill_pending[0] = st->spill_pending[1] = false;
    st->last_spill = 0;
    st->substeps_since_bin = 0;
    d.substeps_since_migration = 0;
    // full EM: only the own planes were stored, the halos come from the neighbours before the next sub-step; the
    // electrostatic cycle recomputes its field from the particles: precalc() (every rank) before the next step()
    d.halos_stale = st->solver == FPIC_SOLVER_YEE;
    st->fields_ready = st->solver == FPIC_SOLVER_POISSON_FFT ? false : hd.fields_ready != 0;
    return FPIC_OK;
}

} // namespace

int save_checkpoint(fpic_handle* h, const char* path)
{
    State* st = h->es;
    if (int rc = em_close_any(h)) return rc;    // (the file holds B of the integer time; a rank forms it from what it holds)
    if (st->dom) return save_rank_checkpoint(h, path);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    BoxFile bf{ std::fopen(path, "wb") };
    if (!bf.f) return fail(h, FPIC_ERR_STATE, "cannot open %s for writing", path);
    BoxCheckpointHeader hd{};
    std::memcpy(hd.magic, "FPICBOX1", 8);
    hd.version = kCheckpointVersion; hd.precision = static_cast<uint32_t>(h->prec); hd.solver = static_cast<uint32_t>(st->solver);
    hd.nspecies = static_cast<uint32_t>(st->sp.size());
    hd.nx = st->nx; hd.ny = st->ny; hd.nz = st->nz; hd.fields_ready = st->fields_ready ? 1 : 0;
    for (int a = 0; a < 3; ++a) hd.B0[a] = st->B0[a];
    hd.spec = h->spec;
    if (std::fwrite(&hd, sizeof hd, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    for (const Species& s : st->sp) {
        const BoxCheckpointSpecies bs{ s.n, s.mass, s.charge };
        if (std::fwrite(&bs, sizeof bs, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint write failed");
    }
    if (int rc = h->prec == FPIC_F32 ? checkpoint_particles<float>(h, bf.f, true) : checkpoint_particles<double>(h, bf.f, true)) return rc;
    return checkpoint_arrays(h, bf.f, true);
}

int load_checkpoint(fpic_handle* h, const char* path)
{
    State* st = h->es;
    if (st->dom) return load_rank_checkpoint(h, path);
    BoxFile bf{ std::fopen(path, "rb") };
    if (!bf.f) return fail(h, FPIC_ERR_STATE, "cannot open %s", path);
    BoxCheckpointHeader hd{};
    // (magic and version are the first twelve bytes whatever the rest of the header looked like when the file was written)
    if (std::fread(&hd, 12, 1, bf.f) != 1 || std::memcmp(hd.magic, "FPICBOX1", 8) != 0)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is not a checkpoint of a box", path);
    if (hd.version != kCheckpointVersion && hd.version != 1) // (version 1: the same layout, written before the number was raised)
        return fail(h, FPIC_ERR_INVALID_ARG, "%s is a box checkpoint of format version %u; this library reads version %u (the header embeds fpic_spec of ABI %d)", path, hd.version,
                    kCheckpointVersion, FPIC_ABI_VERSION);
    if (std::fread(reinterpret_cast<char*>(&hd) + 12, sizeof hd - 12, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
    if (static_cast<int>(hd.precision) != h->prec || static_cast<int>(hd.solver) != st->solver || hd.nspecies != st->sp.size() || hd.nx != st->nx || hd.ny != st->ny ||
        hd.nz != st->nz)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint holds %u species on %d x %d x %d, precision %u, solver %u; the pusher was made for %zu on %d x %d x %d, precision %d, solver %d",
                    hd.nspecies, hd.nx, hd.ny, hd.nz, hd.precision, hd.solver, st->sp.size(), st->nx, st->ny, st->nz, h->prec, st->solver);
    if (hd.spec.radius != h->spec.radius || hd.spec.length_y != h->spec.length_y || hd.spec.height != h->spec.height || hd.spec.dt != h->spec.dt ||
        hd.spec.macro_weight != h->spec.macro_weight)
        return fail(h, FPIC_ERR_INVALID_ARG, ".spec <- checkpoint was written with different lengths / dt / macro_weight");
    const size_t t = h->prec == FPIC_F32 ? 4 : 8;
    unsigned long long want = sizeof hd + hd.nspecies * sizeof(BoxCheckpointSpecies);
    for (size_t k = 0; k < st->sp.size(); ++k) {
        BoxCheckpointSpecies bs{};
        if (std::fread(&bs, sizeof bs, 1, bf.f) != 1) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated");
        if (bs.n != st->sp[k].n || bs.mass != st->sp[k].mass || bs.charge != st->sp[k].charge)
            return fail(h, FPIC_ERR_INVALID_ARG, ".species <- species %zu of the checkpoint (%llu particles, mass %g, charge %g) is not the pusher's", k,
                        static_cast<unsigned long long>(bs.n), bs.mass, bs.charge);
        want += 6ull * bs.n * t;
    }
    for (const auto& a : checkpoint_fields(h)) want += a.second;
    // the whole payload must be there before any device state is touched
    const long at = std::ftell(bf.f);
    if (at < 0 || std::fseek(bf.f, 0, SEEK_END) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    const long long have = std::ftell(bf.f);
    if (have < 0 || static_cast<unsigned long long>(have) < want) return fail(h, FPIC_ERR_STATE, "checkpoint is truncated: %lld bytes, %llu expected", have, want);
    if (std::fseek(bf.f, at, SEEK_SET) != 0) return fail(h, FPIC_ERR_STATE, "cannot seek in %s", path);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (Species& s : st->sp) { // the arrays are about to hold the caller's order: bins and census are void
        s.binned = s.census_fresh = s.rebin_pending = s.chunk_census_fresh = false;
        s.tail_first = s.tail_count = s.n_after = 0;
        s.ids_identity = true; // (ckpt_scatter_kernel writes slot = index)
    }
    st->spill_pending[0] = st->spill_pending[1] = false;
    st->last_spill = 0;
    st->substeps_since_bin = 0;
    st->fields_ready = false;
    if (int rc = h->prec == FPIC_F32 ? checkpoint_particles<float>(h, bf.f, false) : checkpoint_particles<double>(h, bf.f, false)) return rc;
    if (int rc = checkpoint_arrays(h, bf.f, false)) return rc;
    for (int a = 0; a < 3; ++a) st->B0[a] = hd.B0[a];
    st->fields_ready = hd.fields_ready != 0;
    st->em_open = false;    // (the file's B is B of the integer time)
    return FPIC_OK;
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

struct Xfer {
    int to, from;           // ranks
    const void* send;
    size_t send_bytes;
    void* recv;
    size_t recv_bytes;
    int tag;                // a message meets the receive of its destination that names the sender and carries the same tag
};
constexpr int kMigSpecies = 16;                    // species a decomposition can migrate (their counter blocks)
constexpr size_t kMigWords = 8 * (kMigSpecies + 1);  // words of Domain::counts_dev / counts_host
enum Exchange { X_GHOST = 0, X_MIG_COUNTS, X_MIG_PAYLOAD, X_TRANSPOSE, X_TRANSPOSE_BACK, X_PHI, X_EM_J, X_EM_E, X_EM_B };

// A rank of a full-EM decomposition may run the chained lattice step (em_chain_tiled_kernel) when its slab can give H + 1
// planes to a neighbour's halo and the planes it forms the half-time B on, z0 - H .. z0 + nzl + H - 1, are distinct planes
// of the periodic lattice.  The same for every rank (nzl, H and nz are).
bool em_deep_halo(const State* st)
{
    const Domain* d = st->dom;
    return d && d->world > 1 && st->solver == FPIC_SOLVER_YEE && d->H + 1 <= d->nzl && d->nzl + 2 * d->H + 1 <= st->nz;
}

// The messages of one exchange, in an order every rank shares: [0] goes to the slab below and is met there by
// what arrives from above, [1] goes up and is met by what arrives from below.  (RCCL matches the sends and
// receives of a pair of ranks in the order they are issued; with two ranks both messages have the same peer.)
template <typename T>
void dom_xfers(fpic_handle* h, int which, std::vector<Xfer>& out)
{
    State* st = h->es;
    Domain& d = *st->dom;
    const int down = (d.rank + d.world - 1) % d.world, up = (d.rank + 1) % d.world;
    const size_t plane = static_cast<size_t>(st->nx) * st->ny;
    out.clear();
    if (which == X_GHOST) {
        const int lo = (d.z0 - d.G + st->nz) % st->nz, hi = (d.z0 + d.nzl) % st->nz;
        out.push_back({ down, up, st->rho_fixed + lp(st, lo) * plane, d.G * plane * 8, d.ghost_recv[0], d.G * plane * 8, 0 });
        out.push_back({ up, down, st->rho_fixed + lp(st, hi) * plane, (d.G + 1) * plane * 8, d.ghost_recv[1], (d.G + 1) * plane * 8, 1 });
    } else if (which == X_MIG_COUNTS) { // every species' two counts in one exchange
        for (size_t sp = 0; sp < st->sp.size(); ++sp) {
            unsigned* c = d.counts_dev + 8 * sp;
            out.push_back({ down, up, c + 0, 4, c + 4, 4, static_cast<int>(2 * sp) });
            out.push_back({ up, down, c + 1, 4, c + 5, 4, static_cast<int>(2 * sp + 1) });
        }
    } else if (which == X_MIG_PAYLOAD) {
        const size_t rec = sizeof(MigRecord<T>);
        const unsigned* c = d.counts_host + 8 * d.mig_sp;
        out.push_back({ down, up, d.mig_send[0], c[0] * rec, d.mig_recv[0], c[4] * rec, 0 });
        out.push_back({ up, down, d.mig_send[1], c[1] * rec, d.mig_recv[1], c[5] * rec, 1 });
    } else if (which == X_TRANSPOSE || which == X_TRANSPOSE_BACK) {
        // all-to-all of equal chunks: chunk q of the send side goes to rank q and lands there as chunk `rank`
        const size_t chunk = static_cast<size_t>(d.nzl) * d.nyl * row_pitch<T>(st) * 2 * sizeof(T);
        const char* src = static_cast<const char*>(which == X_TRANSPOSE ? d.xbuf : d.hatB);
        char* dst = static_cast<char*>(which == X_TRANSPOSE ? d.hatB : d.xbuf);
        for (int q = 0; q < d.world; ++q) out.push_back({ q, q, src + q * chunk, chunk, dst + q * chunk, chunk, 0 });
    } else if (which == X_EM_J) {
        // the current a rank's particles left on its H ghost planes below / above goes to the slab that owns them
        const int lo = (d.z0 - d.H + st->nz) % st->nz, hi = (d.z0 + d.nzl) % st->nz;
        const size_t bytes = static_cast<size_t>(d.H) * plane * 3 * sizeof(long long);
        out.push_back({ down, up, st->Jfix + 3 * lp(st, lo) * plane, bytes, d.j_recv[0], bytes, 0 });
        out.push_back({ up, down, st->Jfix + 3 * lp(st, hi) * plane, bytes, d.j_recv[1], bytes, 1 });
    } else if (which == X_EM_E || which == X_EM_B) {
        // halo copy of a lattice field: my first H planes are the lower neighbour's upper halo, my last H planes the
        // upper neighbour's lower halo; what arrives lands in my halo planes in place.  E goes one plane deeper into the
        // UPPER halo (the last plane a rank holds, z0 + nzl + H): the chained lattice step forms the half-time B of the
        // top halo plane from the E one plane above it (dom_em_substep).
        T* f = static_cast<T*>(which == X_EM_E ? st->Ey : st->By);
        const size_t bytes = static_cast<size_t>(d.H) * plane * 4 * sizeof(T);
        const size_t deep = static_cast<size_t>(which == X_EM_E && em_deep_halo(st) ? d.H + 1 : d.H) * plane * 4 * sizeof(T);
        const int above = (d.z0 + d.nzl) % st->nz, below = (d.z0 - d.H + st->nz) % st->nz;
        out.push_back({ down, up, f + 4 * lp(st, d.z0) * plane, deep, f + 4 * lp(st, above) * plane, deep, 0 });
        out.push_back({ up, down, f + 4 * lp(st, d.z0 + d.nzl - d.H) * plane, bytes, f + 4 * lp(st, below) * plane, bytes, 1 });
    } else { // X_PHI: the potential on the planes the gradient of my slab and its ghost planes needs
        T* phi = static_cast<T*>(st->phi);
        const int above = (d.z0 + d.nzl) % st->nz, below = (d.z0 - d.phi_below + st->nz) % st->nz;
        const size_t na = d.phi_above * plane * sizeof(T), nb = d.phi_below * plane * sizeof(T);
        out.push_back({ down, up, phi + lp(st, d.z0) * plane, na, phi + lp(st, above) * plane, na, 0 });
        out.push_back({ up, down, phi + lp(st, d.z0 + d.nzl - d.phi_below) * plane, nb, phi + lp(st, below) * plane, nb, 1 });
    }
}

// One rank per process over RCCL (hs.size() == 1), or every rank of a group inside this process (the
// in-process stand-in for the exchange that lets one GPU run and test an N-rank decomposition).
struct Ranks {
    std::vector<fpic_handle*> hs;
    bool rccl = false;
};

template <typename T>
int exchange(Ranks& rk, int which, bool on_comm_stream = false)
{
    if (rk.rccl) {
        fpic_handle* h = rk.hs[0];
        const fdyn::Rccl& rc = fdyn::rccl();
        hipStream_t stream = on_comm_stream && h->es->dom->comm_stream ? h->es->dom->comm_stream : h->stream;
        std::vector<Xfer> x;
        dom_xfers<T>(h, which, x);
        if (int e = fcomm::check(h, rc.GroupStart(), "ncclGroupStart")) return e;
        const int me = h->comm->rank;
        int err = FPIC_OK; // (a group once opened is always closed: an error must not leave the communicator inside it)
        for (const Xfer& m : x) {
            if (m.to == me && m.from == me) continue; // to myself: a copy, below
            if (m.send_bytes && !err) err = fcomm::check(h, rc.Send(m.send, m.send_bytes, ncclChar, m.to, h->comm->nccl, stream), "ncclSend");
            if (m.recv_bytes && !err) err = fcomm::check(h, rc.Recv(m.recv, m.recv_bytes, ncclChar, m.from, h->comm->nccl, stream), "ncclRecv");
        }
        const std::string first = h->err;
        const int end = fcomm::check(h, rc.GroupEnd(), "ncclGroupEnd");
        if (err) { h->err = first; return err; }
        if (end) return end;
        for (const Xfer& m : x)
            if (m.to == me && m.from == me && m.send_bytes && m.recv != m.send)
                HIP_TRY(h, hipMemcpyAsync(m.recv, m.send, m.send_bytes, hipMemcpyDeviceToDevice, stream));
        return FPIC_OK;
    }
    std::vector<std::vector<Xfer>> all(rk.hs.size());
    for (size_t r = 0; r < rk.hs.size(); ++r) dom_xfers<T>(rk.hs[r], which, all[r]);
    for (size_t r = 0; r < rk.hs.size(); ++r)
        for (const Xfer& m : all[r]) {
            const Xfer* peer = nullptr;
            for (const Xfer& c : all[m.to])
                if (c.from == static_cast<int>(r) && c.tag == m.tag) { peer = &c; break; }
            if (!peer || peer->recv_bytes != m.send_bytes)
                return fail(rk.hs[r], FPIC_ERR_STATE, "decomposition exchange %d: a message of rank %zu to rank %d (%zu bytes, tag %d) has no matching receive", which, r,
                            m.to, m.send_bytes, m.tag);
            if (m.send_bytes && peer->recv != m.send)
                HIP_TRY(rk.hs[r], hipMemcpyAsync(peer->recv, m.send, m.send_bytes, hipMemcpyDeviceToDevice, rk.hs[0]->stream));
        }
    return FPIC_OK;
}

// The exchange that follows may start once everything queued so far on the handle's stream has run (fork), and what is
// queued on the handle's stream after the join waits for it.  RCCL transport only: a group has one queue, where the
// order of submission already is the order of execution.
int comm_fork(Ranks& rk)
{
    if (!rk.rccl) return FPIC_OK;
    fpic_handle* h = rk.hs[0];
    Domain& d = *h->es->dom;
    if (!d.comm_stream) return FPIC_OK;
    HIP_TRY(h, hipEventRecord(d.ev_boundary, h->stream));
    if (!(d.test_fault & 1)) HIP_TRY(h, hipStreamWaitEvent(d.comm_stream, d.ev_boundary, 0));
    return FPIC_OK;
}
int comm_join(Ranks& rk)
{
    if (!rk.rccl) return FPIC_OK;
    fpic_handle* h = rk.hs[0];
    Domain& d = *h->es->dom;
    if (!d.comm_stream) return FPIC_OK;
    HIP_TRY(h, hipEventRecord(d.ev_ghost, d.comm_stream));
    if (!(d.test_fault & 2)) HIP_TRY(h, hipStreamWaitEvent(h->stream, d.ev_ghost, 0));
    return FPIC_OK;
}

// every rank ends up with all owned planes of rho
template <typename T>
int allgather_rho(Ranks& rk)
{
    if (rk.rccl) {
        fpic_handle* h = rk.hs[0];
        State* st = h->es;
        const size_t count = static_cast<size_t>(st->dom->nzl) * st->nx * st->ny;
        T* rho = static_cast<T*>(st->rho);
        return fcomm::check(h, fdyn::rccl().AllGather(rho + st->dom->rank * count, rho, count, sizeof(T) == 4 ? ncclFloat : ncclDouble, h->comm->nccl, h->stream),
                            "ncclAllGather");
    }
    for (fpic_handle* dst : rk.hs)
        for (fpic_handle* src : rk.hs) {
            if (src == dst) continue;
            const State* ss = src->es;
            const size_t count = static_cast<size_t>(ss->dom->nzl) * ss->nx * ss->ny, off = ss->dom->rank * count;
            HIP_TRY(dst, hipMemcpyAsync(static_cast<T*>(dst->es->rho) + off, static_cast<const T*>(ss->rho) + off, count * sizeof(T), hipMemcpyDeviceToDevice,
                                        rk.hs[0]->stream));
        }
    return FPIC_OK;
}

// every rank ends up with every rank's interface planes (and its piece of the (0, 0) mode's line): fes_tri.hpp, step 2
template <typename T>
int allgather_tri(Ranks& rk)
{
    if (rk.rccl) {
        fpic_handle* h = rk.hs[0];
        const Domain& d = *h->es->dom;
        T* buf = static_cast<T*>(d.tri);
        const size_t count = 2 * d.tri_block; // scalars per rank
        return fcomm::check(h, fdyn::rccl().AllGather(buf + d.rank * count, buf, count, sizeof(T) == 4 ? ncclFloat : ncclDouble, h->comm->nccl, h->stream), "ncclAllGather");
    }
    for (fpic_handle* dst : rk.hs)
        for (fpic_handle* src : rk.hs) {
            if (src == dst) continue;
            const Domain& sd = *src->es->dom;
            const size_t count = 2 * sd.tri_block, off = sd.rank * count;
            HIP_TRY(dst, hipMemcpyAsync(static_cast<T*>(dst->es->dom->tri) + off, static_cast<const T*>(sd.tri) + off, count * sizeof(T), hipMemcpyDeviceToDevice, rk.hs[0]->stream));
        }
    return FPIC_OK;
}

// One word agreed by every rank (maximum): an error that only one rank sees must stop them all at the same point of the
// exchange sequence, or the others wait in the next send / receive for ever.
int agree_max(Ranks& rk, unsigned mine_of_rank0, const std::vector<unsigned>& mine, unsigned& out)
{
    if (!rk.rccl) {
        out = 0;
        for (unsigned v : mine) out = std::max(out, v);
        return FPIC_OK;
    }
    fpic_handle* h = rk.hs[0];
    Domain& d = *h->es->dom;
    constexpr size_t W = 8 * kMigSpecies; // (the word after the species' counter blocks)
    d.counts_host[W] = mine_of_rank0;
    HIP_TRY(h, hipMemcpyAsync(d.counts_dev + W, d.counts_host + W, sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
    if (int e = fcomm::check(h, fdyn::rccl().AllReduce(d.counts_dev + W, d.counts_dev + W, 1, ncclUint32, ncclMax, h->comm->nccl, h->stream), "ncclAllReduce")) return e;
    HIP_TRY(h, hipMemcpyAsync(d.counts_host + W, d.counts_dev + W, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    out = d.counts_host[W];
    return FPIC_OK;
}

// particles that have left the slab move to the neighbour that owns them; then every species is re-binned.
// Every rank takes the same path through the exchanges: a leaver that does not fit the message stays where it is (it
// still deposits on the ghost planes) and leaves with the next migration; a rank that cannot hold its arrivals is an
// error that ALL ranks return — agreed ONCE, for every species, before the first particle of any species is touched
// (round 4: a count-only scan of every species first; rounds 2-3 agreed species by species, so a later species' refusal
// came after earlier ones had moved, and cost a blocking all-reduce per species).
template <typename T>
int migrate(Ranks& rk)
{
    const size_t nsp = rk.hs[0]->es->sp.size();
    if (nsp > static_cast<size_t>(kMigSpecies))
        return fail(rk.hs[0], FPIC_ERR_STATE, "migration: %zu species, the decomposition's counters hold %d", nsp, kMigSpecies);
    for (fpic_handle* h : rk.hs) timing_begin(h, KC_SORT);
    // the scan of one species of one rank: its leavers counted (COUNT_ONLY) or packed into the two messages
    auto scan = [&](fpic_handle* h, size_t sp, bool count_only) -> int {
        State* st = h->es;
        Domain& d = *st->dom;
        Species& s = st->sp[sp];
        unsigned* counts = d.counts_dev + 8 * sp;
        HIP_TRY(h, hipMemsetAsync(counts, 0, 4 * sizeof(unsigned), h->stream)); // (what has arrived in words 4, 5 stays)
        // a species whose last push left a census of the current positions is not re-binned by separate passes:
        // the census is corrected for leavers and arrivals and the next push re-bins (and compacts) itself
        const bool riding = s.binned && s.census_fresh && st->solver != FPIC_SOLVER_YEE;
        if (s.n) {
            // a species binned since its last upload is scanned along the slab's faces only: the interior tile layers
            // (interior_layers: the same rule as the two-part push) cannot hold a leaver
            uint32_t lo = 0, hi = 0;
            const bool faces_only = s.binned && interior_layers(st, lo, hi);
            const uint32_t per_layer = static_cast<uint32_t>(st->ntx) * st->nty;
            const unsigned grid = std::min<unsigned>(blocks_for(s.n, 256 * kMigPer), 4096u);
            const uint32_t* ts = faces_only ? s.tile_start2[s.wl] : nullptr;
            if (count_only)
                mig_pack_kernel<T, true><<<grid, 256, 0, h->stream>>>(static_cast<T*>(s.slab[s.cur]), s.n_pad, s.id[s.cur], s.n, st->nz, d.z0, d.nzl, d.G, d.world, nullptr, nullptr,
                                                                     d.mig_cap, counts, nullptr, st->nx, st->ny, st->ntx, st->nty, ts, lo * per_layer, hi * per_layer);
            else
                mig_pack_kernel<T, false><<<grid, 256, 0, h->stream>>>(static_cast<T*>(s.slab[s.cur]), s.n_pad, s.id[s.cur], s.n, st->nz, d.z0, d.nzl, d.G, d.world,
                                                                      static_cast<MigRecord<T>*>(d.mig_send[0]), static_cast<MigRecord<T>*>(d.mig_send[1]), d.mig_cap, counts,
                                                                      riding ? s.tile_count : nullptr, st->nx, st->ny, st->ntx, st->nty, ts, lo * per_layer, hi * per_layer);
        }
        // the message counters counted every leaver; what the messages hold is at most mig_cap records each
        mig_clamp_kernel<<<1, 64, 0, h->stream>>>(counts, d.mig_cap, count_only ? 1 : 0);
        HIP_TRY(h, hipGetLastError());
        return FPIC_OK;
    };
    // 1. every species counted, the counts exchanged, ONE verdict
    for (size_t sp = 0; sp < nsp; ++sp)
        for (fpic_handle* h : rk.hs)
            if (int e = scan(h, sp, true)) return e;
    if (int e = exchange<T>(rk, X_MIG_COUNTS)) return e;
    std::vector<unsigned> verdict(rk.hs.size(), 0u);
    for (size_t r = 0; r < rk.hs.size(); ++r) {
        fpic_handle* h = rk.hs[r];
        Domain& d = *h->es->dom;
        HIP_TRY(h, hipMemcpyAsync(d.counts_host, d.counts_dev, 8 * nsp * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (size_t sp = 0; sp < nsp && !verdict[r]; ++sp) {
            const Species& s = h->es->sp[sp];
            const unsigned* c = d.counts_host + 8 * sp;
            const size_t in = static_cast<size_t>(c[4]) + c[5], out = static_cast<size_t>(c[0]) + c[1];
            if (c[4] > d.mig_cap || c[5] > d.mig_cap) {
                verdict[r] = 2;
                fail(h, FPIC_ERR_STATE, "migration: rank %d would receive %u and %u records of species %zu, its message buffers hold %u", d.rank, c[4], c[5], sp, d.mig_cap);
            } else if (s.n + in > s.n_pad || s.n - out + in > s.cap) {
                verdict[r] = 1;
                fail(h, FPIC_ERR_STATE, "migration: rank %d would hold %zu particles of species %zu, capacity %zu", d.rank, s.n - out + in, sp, s.cap);
            }
        }
    }
    unsigned worst = 0;
    if (int e = agree_max(rk, verdict[0], verdict, worst)) return e;
    if (worst) { // nothing has been touched: every rank returns the error
        int first_bad = -1;
        for (size_t r = 0; r < rk.hs.size(); ++r) {
            fpic_handle* h = rk.hs[r];
            if (verdict[r] && first_bad < 0) first_bad = static_cast<int>(r);
            else if (!verdict[r]) fail(h, FPIC_ERR_STATE, "migration: another rank cannot hold its arrivals; nothing was moved");
            timing_end(h);
        }
        if (first_bad > 0) rk.hs[0]->err = rk.hs[first_bad]->err; // (a group reports through its first member)
        return FPIC_ERR_STATE;
    }
    // 2. species by species: pack (the same scan: the same counts), payload, arrivals, the next bin table
    for (size_t sp = 0; sp < nsp; ++sp) {
        for (fpic_handle* h : rk.hs) {
            if (int e = scan(h, sp, false)) return e;
            Domain& d = *h->es->dom;
            const unsigned* c = d.counts_host + 8 * sp;
            d.mig_sp = static_cast<int>(sp);
            d.lost += c[2];
            d.deferred += c[3];
            d.migrated += c[0] + c[1];
        }
        if (int e = exchange<T>(rk, X_MIG_PAYLOAD)) return e;
        for (fpic_handle* h : rk.hs) {
            State* st = h->es;
            Domain& d = *st->dom;
            Species& s = st->sp[sp];
            const unsigned* c = d.counts_host + 8 * sp;
            const size_t in = static_cast<size_t>(c[4]) + c[5];
            const size_t out = static_cast<size_t>(c[0]) + c[1];
            T* slab = static_cast<T*>(s.slab[s.cur]);
            const bool riding = s.binned && s.census_fresh && st->solver != FPIC_SOLVER_YEE;
            uint32_t* census = riding ? s.tile_count : nullptr;
            if (c[4])
                mig_append_kernel<T><<<blocks_for(c[4]), 256, 0, h->stream>>>(static_cast<const MigRecord<T>*>(d.mig_recv[0]), c[4], slab, s.n_pad, s.id[s.cur], s.n, census, st->nx,
                                                                             st->ny, st->nz, st->ntx, st->nty);
            if (c[5])
                mig_append_kernel<T><<<blocks_for(c[5]), 256, 0, h->stream>>>(static_cast<const MigRecord<T>*>(d.mig_recv[1]), c[5], slab, s.n_pad, s.id[s.cur], s.n + c[4], census,
                                                                             st->nx, st->ny, st->nz, st->ntx, st->nty);
            HIP_TRY(h, hipGetLastError());
            if (riding) {
                // the next bin table from the corrected census; the push that follows skips the dead slots, takes the
                // arrivals from the tail and leaves a compact sorted array in the other set
                const int nw = s.wl ^ 1;
                bin_scan_kernel<<<1, 1024, 0, h->stream>>>(s.tile_count, st->ntiles, s.tile_start2[nw], s.tile_cursor, s.work2[nw], s.nwork2[nw], static_cast<uint32_t>(kChunk3));
                HIP_TRY(h, hipGetLastError());
                s.rebin_pending = true;
                s.tail_first = s.n; s.tail_count = in;
                s.n_after = s.n - out + in;
                if (s.n_after == 0) {
                    // an emptied rank: nothing to push.  It stays "binned" (an empty array is sorted; launch_bin says the
                    // same for n == 0), so that the decision to migrate — which every rank must take alike, or their
                    // exchanges no longer pair up — never depends on one rank's population; later arrivals are binned by
                    // the separate passes of the next migration
                    s.n = 0;
                    s.rebin_pending = s.census_fresh = s.chunk_census_fresh = false;
                    s.binned = true;
                    s.tail_first = s.tail_count = 0;
                }
            } else {
                // the binning runs over the old slots (dead ones skipped) and the arrivals, and leaves a compact array
                const size_t slots = s.n + in;
                s.n = slots;
                if (int e = launch_bin<T>(h, s)) return e;
                s.n = slots - out;
            }
        }
    }
    for (fpic_handle* h : rk.hs) {
        timing_end(h);
        State* st = h->es;
        st->substeps_since_bin = 0;
        st->dom->substeps_since_migration = 0;
        h->sort_passes++;
    }
    return FPIC_OK;
}

// The Poisson solve of a decomposed run without any rank holding the whole spectrum: per rank 2-D real transforms of
// its nzl planes, an all-to-all transposition (each pair of ranks exchanges nzl * nyl * nxh complex values), the
// transforms along z and the k-space factor on the rank's nyl rows of ky, the transposition back, the inverse 2-D
// transforms, and the potential of G + 1 / G + 2 neighbouring planes for the gradient on the slab and its ghost planes.
template <typename T>
int solve_distributed(Ranks& rk)
{
    const fdyn::RocFFT& ff = fdyn::rocfft();
    auto each = [&](auto fn) -> int {
        for (fpic_handle* h : rk.hs)
            if (int e = fn(h)) return e;
        return FPIC_OK;
    };
    auto run_fft = [&](fpic_handle* h, rocfft_plan plan, rocfft_execution_info info, void* in, void* out, const char* what) -> int {
        if (int e = fft_status(h, ff.execution_info_set_stream(info, h->stream), "rocfft_execution_info_set_stream")) return e;
        void* ib[1] = { in };
        void* ob[1] = { out };
        return fft_status(h, ff.execute(plan, ib, ob, info), what);
    };
    const bool own = rk.hs[0]->es->own_fft;
    if (rk.hs[0]->es->dom->interface_solve) {
        // fes_tri.hpp: x and y transforms of the own planes in place, down sweep along z, all-gather of two planes per rank,
        // up sweep, inverse y and x transforms — no transposition, 1/32 of its bytes on the links (nzl = 64)
        if (int e = each([&](fpic_handle* h) -> int {
                State* st = h->es;
                Domain& d = *st->dom;
                const int pitch = static_cast<int>(row_pitch<T>(st));
                const size_t plane = static_cast<size_t>(st->nx) * st->ny;
                timing_begin(h, KC_SOLVE);
                const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz), dz = st->lz / st->nz;
                const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv);
                T* hat = static_cast<T*>(d.hatA);
                if (int e2 = fft_x_forward<T>(h, st->rho_fixed + lp(st, d.z0) * plane, nullptr, scale, static_cast<size_t>(d.nzl) * st->ny, hat)) return e2;
                if (int e2 = fft_columns<T, 0>(h, hat, static_cast<size_t>(st->ny) * pitch, pitch, d.nzl, st->ny)) return e2;
                const festri::Slab sl{ d.nzl, st->ny, st->nx / 2 + 1, pitch };
                T* mine = static_cast<T*>(d.tri) + 2 * (static_cast<size_t>(d.rank) * d.tri_block);
                festri::tri_down_kernel<T><<<blocks_for(static_cast<size_t>(st->ny) * pitch), 256, 0, h->stream>>>(
                    hat, sl, st->k2[0], st->k2[1], dz * dz, dz * dz / (kEps0 * static_cast<double>(st->nx) * st->ny), mine, mine + 2 * (2 * static_cast<size_t>(st->ny) * pitch));
                HIP_TRY(h, hipGetLastError());
                return FPIC_OK;
            })) return e;
        if (int e = allgather_tri<T>(rk)) return e;
        if (int e = each([&](fpic_handle* h) -> int {
                State* st = h->es;
                Domain& d = *st->dom;
                const int pitch = static_cast<int>(row_pitch<T>(st));
                const size_t plane = static_cast<size_t>(st->nx) * st->ny;
                const double dz = st->lz / st->nz;
                T* hat = static_cast<T*>(d.hatA);
                const festri::Slab sl{ d.nzl, st->ny, st->nx / 2 + 1, pitch };
                festri::tri_up_kernel<T><<<blocks_for(static_cast<size_t>(st->ny) * pitch), 256, 0, h->stream>>>(
                    hat, sl, st->k2[0], st->k2[1], dz * dz, static_cast<const T*>(d.tri), d.tri_block, d.world, d.rank);
                HIP_TRY(h, hipGetLastError());
                festri::tri_zero_line_kernel<T><<<1, 1024, 0, h->stream>>>(hat, sl, static_cast<const T*>(d.tri), d.tri_block, d.world, d.rank);
                HIP_TRY(h, hipGetLastError());
                if (int e2 = fft_columns<T, 1>(h, hat, static_cast<size_t>(st->ny) * pitch, pitch, d.nzl, st->ny)) return e2;
                return fft_x_inverse<T>(h, hat, static_cast<size_t>(d.nzl) * st->ny, static_cast<T*>(st->phi) + lp(st, d.z0) * plane);
            })) return e;
    } else {
    if (int e = each([&](fpic_handle* h) -> int {
            State* st = h->es;
            Domain& d = *st->dom;
            const int nxh = static_cast<int>(row_pitch<T>(st)); // (the rows' pitch: nx / 2 + 1 with rocFFT)
            const size_t plane = static_cast<size_t>(st->nx) * st->ny;
            timing_begin(h, KC_SOLVE);
            if (own) { // x pass straight from the integer grid of the own planes, then the y pass
                const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
                const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv);
                if (int e2 = fft_x_forward<T>(h, st->rho_fixed + lp(st, d.z0) * plane, nullptr, scale, static_cast<size_t>(d.nzl) * st->ny, static_cast<T*>(d.hatA))) return e2;
                // (the y pass stores straight into the all-to-all's send buffer: no pack sweep)
                return fft_columns<T, 0>(h, static_cast<T*>(d.hatA), static_cast<size_t>(st->ny) * nxh, nxh, d.nzl, st->ny, 0, static_cast<T*>(d.xbuf), d.nyl, d.nzl);
            } else if (int e2 = run_fft(h, d.p2f, d.i2f, static_cast<T*>(st->rho) + lp(st, d.z0) * plane, d.hatA, "rocfft_execute (2-D forward)")) {
                return e2;
            }
            const size_t total = static_cast<size_t>(nxh) * st->ny * d.nzl;
            transpose_pack_kernel<T><<<blocks_for(total), 256, 0, h->stream>>>(static_cast<const T*>(d.hatA), nxh, st->ny, d.nzl, d.nyl, static_cast<T*>(d.xbuf));
            HIP_TRY(h, hipGetLastError());
            return FPIC_OK;
        })) return e;
    if (int e = exchange<T>(rk, X_TRANSPOSE)) return e;
    if (int e = each([&](fpic_handle* h) -> int {
            State* st = h->es;
            Domain& d = *st->dom;
            const int nxh = static_cast<int>(row_pitch<T>(st)); // (the rows' pitch: nx / 2 + 1 with rocFFT)
            if (own) // the whole z direction in one sweep over hatB [nz][nyl][nxh]: forward, k-space factor, inverse
                return fft_columns<T, 2>(h, static_cast<T*>(d.hatB), nxh, static_cast<size_t>(d.nyl) * nxh, d.nyl, st->nz, d.rank * d.nyl);
            // hatB [nz][nyl][nxh] -> hatZ [nyl * nxh][nz], contiguous transforms along z, the k-space factor, and back
            const int cols = d.nyl * nxh;
            const dim3 gf((cols + 31) / 32, (st->nz + 31) / 32), gb((st->nz + 31) / 32, (cols + 31) / 32);
            transpose_complex_kernel<T><<<gf, 256, 0, h->stream>>>(static_cast<const T*>(d.hatB), static_cast<T*>(d.hatZ), st->nz, cols);
            HIP_TRY(h, hipGetLastError());
            if (int e2 = run_fft(h, d.pzf, d.izf, d.hatZ, d.hatZ, "rocfft_execute (z forward)")) return e2;
            const size_t modes = static_cast<size_t>(nxh) * d.nyl * st->nz;
            kspace_zmajor_kernel<T><<<blocks_for(modes), 256, 0, h->stream>>>(static_cast<T*>(d.hatZ), nxh, d.nyl, st->nz, d.rank * d.nyl, st->k2[0], st->k2[1], st->k2[2],
                                                                             1.0 / (kEps0 * static_cast<double>(st->nodes)));
            HIP_TRY(h, hipGetLastError());
            if (int e2 = run_fft(h, d.pzi, d.izi, d.hatZ, d.hatZ, "rocfft_execute (z inverse)")) return e2;
            transpose_complex_kernel<T><<<gb, 256, 0, h->stream>>>(static_cast<const T*>(d.hatZ), static_cast<T*>(d.hatB), cols, st->nz);
            HIP_TRY(h, hipGetLastError());
            return FPIC_OK;
        })) return e;
    if (int e = exchange<T>(rk, X_TRANSPOSE_BACK)) return e;
    if (int e = each([&](fpic_handle* h) -> int {
            State* st = h->es;
            Domain& d = *st->dom;
            const int nxh = static_cast<int>(row_pitch<T>(st)); // (the rows' pitch: nx / 2 + 1 with rocFFT)
            const size_t plane = static_cast<size_t>(st->nx) * st->ny;
            const size_t total = static_cast<size_t>(nxh) * st->ny * d.nzl;
            if (own) { // (the y pass loads straight from the all-to-all's receive buffer: no unpack sweep)
                if (int e2 = fft_columns<T, 1>(h, static_cast<T*>(d.hatA), static_cast<size_t>(st->ny) * nxh, nxh, d.nzl, st->ny, 0, static_cast<T*>(d.xbuf), d.nyl, d.nzl)) return e2;
                return fft_x_inverse<T>(h, static_cast<const T*>(d.hatA), static_cast<size_t>(d.nzl) * st->ny, static_cast<T*>(st->phi) + lp(st, d.z0) * plane);
            }
            transpose_unpack_kernel<T><<<blocks_for(total), 256, 0, h->stream>>>(static_cast<const T*>(d.xbuf), nxh, st->ny, d.nzl, d.nyl, static_cast<T*>(d.hatA));
            HIP_TRY(h, hipGetLastError());
            return run_fft(h, d.p2i, d.i2i, d.hatA, static_cast<T*>(st->phi) + lp(st, d.z0) * plane, "rocfft_execute (2-D inverse)");
        })) return e;
    }
    // The potential's ghost planes travel (RCCL: on the communicator's stream) while the gradient of the planes that need
    // none of them is formed: a plane's gradient reads its two neighbours, so the slab's inner nzl - 2 planes are free.
    bool split = rk.hs[0]->es->solver != FPIC_SOLVER_YEE;
    for (fpic_handle* h : rk.hs) split &= h->es->dom->overlap && h->es->dom->nzl >= 3;
    auto gradient = [&](fpic_handle* h, int first, int count) -> int {
        State* st = h->es;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny;
        gradient_planes_kernel<T><<<blocks_for(plane * count), 256, 0, h->stream>>>(
            static_cast<const T*>(st->phi), st->nx, st->ny, st->nz, first, count, static_cast<T>(1.0 / (2.0 * (st->lx / st->nx))),
            static_cast<T>(1.0 / (2.0 * (st->ly / st->ny))), static_cast<T>(1.0 / (2.0 * (st->lz / st->nz))), static_cast<T*>(st->E4), held_of(st));
        HIP_TRY(h, hipGetLastError());
        return FPIC_OK;
    };
    if (split) {
        if (int e = comm_fork(rk)) return e;
        if (int e = exchange<T>(rk, X_PHI, /*on_comm_stream=*/true)) return e;
        if (int e = each([&](fpic_handle* h) -> int { return gradient(h, h->es->dom->z0 + 1, h->es->dom->nzl - 2); })) return e;
        if (int e = comm_join(rk)) return e;
        return each([&](fpic_handle* h) -> int {
            const Domain& d = *h->es->dom;
            if (int e = gradient(h, d.z0 - d.G, d.G + 1)) return e;           // ghost planes below and the slab's first plane
            if (int e = gradient(h, d.z0 + d.nzl - 1, d.G + 2)) return e;     // the slab's last plane and the ghost planes above
            timing_end(h);
            h->solve_launches++;
            return FPIC_OK;
        });
    }
    if (int e = exchange<T>(rk, X_PHI)) return e;
    return each([&](fpic_handle* h) -> int {
        State* st = h->es;
        Domain& d = *st->dom;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny;
        if (st->solver == FPIC_SOLVER_YEE) { // the initial E on the edges of the slab and of its H halo planes on either side
            const int count = d.nzl + 2 * d.H;
            em_edge_gradient_kernel<T><<<blocks_for(plane * count), 256, 0, h->stream>>>(
                static_cast<const T*>(st->phi), st->nx, st->ny, st->nz, static_cast<T>(1.0 / (st->lx / st->nx)), static_cast<T>(1.0 / (st->ly / st->ny)),
                static_cast<T>(1.0 / (st->lz / st->nz)), static_cast<T*>(st->Ey), ((d.z0 - d.H) % st->nz + st->nz) % st->nz, count, held_of(st));
            HIP_TRY(h, hipGetLastError());
        } else if (int e = gradient(h, d.z0 - d.G, d.nzl + 2 * d.G + 1)) {
            return e;
        }
        timing_end(h);
        h->solve_launches++;
        return FPIC_OK;
    });
}

template <typename T>
int dom_fields(Ranks& rk, bool ghost_exchanged)
{
    const bool multi = rk.hs[0]->es->dom->world > 1;
    if (multi && !ghost_exchanged)
        if (int e = exchange<T>(rk, X_GHOST)) return e;
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        Domain& d = *st->dom;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny;
        timing_begin(h, KC_SOLVE);
        if (multi) {
            // from above: the upper neighbour's lower ghost planes = my top G planes; from below: its G + 1 upper ghost planes = my first ones
            ghost_add_kernel<<<blocks_for(d.G * plane), 256, 0, h->stream>>>(st->rho_fixed + lp(st, d.z0 + d.nzl - d.G) * plane, d.ghost_recv[0], d.G * plane);
            ghost_add_kernel<<<blocks_for((d.G + 1) * plane), 256, 0, h->stream>>>(st->rho_fixed + lp(st, d.z0) * plane, d.ghost_recv[1], (d.G + 1) * plane);
        }
        // the own planes as T: what the replicated solve gathers and what rocFFT's 2-D transforms read; the library's own
        // x pass reads the integer grid itself
        const bool needs_rho = multi && !(d.distributed && st->own_fft);
        if (needs_rho) {
            const double dv = (st->lx / st->nx) * (st->ly / st->ny) * (st->lz / st->nz);
            const double scale = h->spec.particle_charge * st->W / (4398046511104.0 * dv);
            const size_t own = d.nzl * plane, off = lp(st, d.z0) * plane;
            rho_real_kernel<T><<<blocks_for(own), 256, 0, h->stream>>>(st->rho_fixed + off, own, scale, static_cast<T*>(st->rho) + off);
            st->rho_fresh = true;
        } else if (multi) {
            st->rho_fresh = false;
        }
        HIP_TRY(h, hipGetLastError());
        timing_end(h);
    }
    if (multi && rk.hs[0]->es->dom->distributed) return solve_distributed<T>(rk);
    if (multi)
        if (int e = allgather_rho<T>(rk)) return e;
    for (fpic_handle* h : rk.hs)
        if (int e = launch_solve<T>(h, /*convert=*/!multi)) return e; // (a world of one is one handle's solve)
    return FPIC_OK;
}

template <typename T>
int dom_em_after_precalc(Ranks& rk);

template <typename T>
int dom_fields(Ranks& rk, bool ghost_exchanged);

template <typename T>
int dom_precalc(Ranks& rk)
{
    for (fpic_handle* h : rk.hs) {
        // a "decomposition" of one rank (bench.py's strong_c4 at N = 1) bins a large fresh population before its first deposit
        // like an undecomposed handle (precalc()); the ranks of a real decomposition do not: whether a rank is binned decides
        // whether its first sub-step migrates, and that decision must not depend on one rank's population
        if (h->es->dom->world == 1 && h->es->solver != FPIC_SOLVER_YEE) {
            bool bin_first = false;
            for (const Species& sp : h->es->sp) bin_first |= !sp.binned && sp.n >= h->two_level_min;
            if (bin_first)
                if (int e = bin_all<T>(h, false)) return e;
        }
        if (int e = deposit_cycle<T, true>(h)) return e;
        h->deposit_launches++;
    }
    if (int e = dom_fields<T>(rk, false)) return e;
    if (rk.hs[0]->es->solver == FPIC_SOLVER_YEE)
        if (int e = dom_em_after_precalc<T>(rk)) return e;
    for (fpic_handle* h : rk.hs) h->es->fields_ready = true;
    return FPIC_OK;
}

// density() on the ranks of a decomposed full-EM run (the cycle deposits currents; the reference's frame loop still calls
// density() every frame, fusionsim.js:174): every rank deposits the charge of the particles it holds on its own planes
// and its ghost planes, the ghost planes travel to the slabs that own them and are added there (exact: int64).  After it
// FPIC_F3_RHO_FIXED is complete on every rank's own planes.  Collective: every rank calls it.
template <typename T>
int dom_density(Ranks& rk)
{
    for (fpic_handle* h : rk.hs) {
        if (int e = deposit_cycle<T, true>(h)) return e;
        h->deposit_launches++;
    }
    if (rk.hs[0]->es->dom->world < 2) return FPIC_OK;
    if (int e = exchange<T>(rk, X_GHOST)) return e;
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        Domain& d = *st->dom;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny;
        ghost_add_kernel<<<blocks_for(d.G * plane), 256, 0, h->stream>>>(st->rho_fixed + lp(st, d.z0 + d.nzl - d.G) * plane, d.ghost_recv[0], d.G * plane);
        ghost_add_kernel<<<blocks_for((d.G + 1) * plane), 256, 0, h->stream>>>(st->rho_fixed + lp(st, d.z0) * plane, d.ghost_recv[1], (d.G + 1) * plane);
        HIP_TRY(h, hipGetLastError());
        st->rho_fresh = false;
    }
    return FPIC_OK;
}

// ---- the full-EM cycle of a decomposition.  Every rank keeps the lattice fields of its slab and of H = G + 2 halo
// planes on each side current: after the E update and after the second B half step the boundary planes are copied to
// the neighbours (X_EM_E, X_EM_B); the current of a sub-step is completed on the owned planes by adding what the
// neighbours' particles left on their ghost planes (X_EM_J, exact: int64).  Plane for plane the arithmetic is the one
// handle's, so the fields, the currents and the particles are bit-identical to an undecomposed run.
template <typename T>
int dom_em_after_precalc(Ranks& rk)
{
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        fill4_kernel<T><<<blocks_for(held_nodes(st)), 256, 0, h->stream>>>(static_cast<T*>(st->By), held_nodes(st), static_cast<T>(st->B0[0]), static_cast<T>(st->B0[1]),
                                                                         static_cast<T>(st->B0[2]));
        HIP_TRY(h, hipGetLastError());
        st->em_open = false;                   // (both lattice fields are set afresh)
        if (int e = em_nodes<T>(h)) return e; // (every rank has solved the whole grid: its E is valid everywhere)
    }
    // (with the decomposed solve E comes from the potential on the slab and its H halo planes; the one plane above them that
    // the chained lattice step reads arrives with a halo copy — collective, as precalc() is)
    if (rk.hs[0]->es->dom->world > 1 && em_deep_halo(rk.hs[0]->es))
        if (int e = exchange<T>(rk, X_EM_E)) return e;
    return FPIC_OK;
}

template <typename T>
int dom_em_substep(Ranks& rk)
{
    State* s0 = rk.hs[0]->es;
    const bool multi = s0->dom->world > 1;
    bool unbinned = false;
    for (fpic_handle* h : rk.hs)
        for (const Species& s : h->es->sp) unbinned |= !s.binned;
    if (multi) {
        if (unbinned || s0->dom->substeps_since_migration >= s0->dom->migrate_every)
            if (int e = migrate<T>(rk)) return e;
    } else if (unbinned || s0->substeps_since_bin >= 64) {
        for (fpic_handle* h : rk.hs)
            if (int e = bin_all<T>(h, true)) return e;
    }
    bool split = multi;
    for (fpic_handle* h : rk.hs) split &= can_split(h->es);
    if (multi && s0->dom->halos_stale) { // fields restored from the ranks' checkpoints: own planes only
        if (int e = exchange<T>(rk, X_EM_E)) return e;
        if (int e = exchange<T>(rk, X_EM_B)) return e;
        for (fpic_handle* h : rk.hs) h->es->dom->halos_stale = false;
    }
    // The chained lattice step on the ranks of a decomposition (round 4; em_substep has the undecomposed form).  A rank keeps
    // the half-time B on the planes z0 - H .. z0 + nzl + H - 1 — its slab and its halos — and forms it there ITSELF, from the
    // E halo it receives (one plane deeper above: X_EM_E), with the arithmetic its neighbours use on their own planes: the
    // second B half step, the node centring and the next first half step are one sweep, and the halo copy of B is gone
    // (half the lattice halo bytes).  B of the integer time is formed when somebody reads or replaces it (em_close), plane
    // for plane on whatever the rank holds — no exchange, so one rank may close and reopen without the others.
    // FPIC_EM_CHAIN=0 keeps the four sweeps and both halo copies.
    // The switch is read ONCE, by fpic_domain_init, and agreed by the ranks before the first sub-step: it changes the exchange
    // sequence (a chained rank skips X_EM_B and sends the deeper E halo), so ranks that disagreed would wait for ever.
    if (multi && !s0->dom->em_chain_agreed) {
        std::vector<unsigned> off(rk.hs.size());
        for (size_t r = 0; r < rk.hs.size(); ++r) off[r] = rk.hs[r]->es->dom->em_chain ? 0u : 1u;
        unsigned any_off = 0;
        if (int e = agree_max(rk, off[0], off, any_off)) return e;
        for (fpic_handle* h : rk.hs) {
            if (any_off) h->es->dom->em_chain = false;   // one rank without it: nobody chains
            h->es->dom->em_chain_agreed = true;
        }
    }
    const bool chain = multi && em_deep_halo(s0) && s0->dom->em_chain;
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        const Domain& d = *st->dom;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny;
        HIP_TRY(h, hipMemsetAsync(st->spilled, 0, sizeof(unsigned long long), h->stream));
        if (!chain && st->em_open)      // (the switch was turned off between two sub-steps)
            if (int e = em_close<T>(h)) return e;
        if (chain && (!st->Bh[0] || !st->Bh[1]))
            if (int e = alloc_half_time<T>(h, held_nodes(st))) return e;
        // node-centred fields where this rank's particles can be: cells [z0 - G, z0 + nzl + G) -> nodes one further up
        if (chain && st->em_open) {
            timing_begin(h, KC_SOLVE);
            const EmCoef<T> co(h);
            if (int e = em_chain_launch<T>(h, co, d.z0 - d.G - 1, d.nzl + 2 * d.G + 3, true)) return e;
            timing_end(h);
            st->bh_cur ^= 1;
        } else if (int e = multi ? em_nodes<T>(h, d.z0 - d.G - 1, std::min(st->nz, d.nzl + 2 * d.G + 3)) : em_nodes<T>(h)) {
            return e;
        }
        timing_begin(h, KC_PUSH);
        if (multi) { // the planes the slab's particles can deposit on
            if (int e = zero_planes(h, st->Jfix, 3 * plane * sizeof(long long), d.z0 - d.H, d.nzl + 2 * d.H)) return e;
        } else {
            HIP_TRY(h, hipMemsetAsync(st->Jfix, 0, st->nodes * 3 * sizeof(long long), h->stream));
        }
        if (int e = em_push_all<T>(h, split ? 1 : 0)) return e;
        if (!split) timing_end(h);
    }
    if (split) { // the current's ghost planes travel while the interior of the slab is pushed (see dom_substep)
        if (int e = comm_fork(rk)) return e;
        if (int e = exchange<T>(rk, X_EM_J, /*on_comm_stream=*/true)) return e;
        for (fpic_handle* h : rk.hs) {
            if (int e = em_push_all<T>(h, 2)) return e;
            timing_end(h);
        }
        if (int e = comm_join(rk)) return e;
    } else if (multi) {
        if (int e = exchange<T>(rk, X_EM_J)) return e;
    }
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        const Domain& d = *st->dom;
        const size_t plane = static_cast<size_t>(st->nx) * st->ny, count = 3 * d.H * plane;
        timing_begin(h, KC_SOLVE);
        const EmCoef<T> co(h);
        if (multi) {
            // from above: the upper neighbour's lower ghost planes = my last H planes; from below: my first H planes
            ghost_add_kernel<<<blocks_for(count), 256, 0, h->stream>>>(st->Jfix + 3 * lp(st, d.z0 + d.nzl - d.H) * plane, d.j_recv[0], count);
            ghost_add_kernel<<<blocks_for(count), 256, 0, h->stream>>>(st->Jfix + 3 * lp(st, d.z0) * plane, d.j_recv[1], count);
            HIP_TRY(h, hipGetLastError());
            if (chain) {
                if (!st->em_open) { // from B at the integer time: its first half step on every plane the rank reads it on
                    st->bh_cur = 0;
                    if (int e = em_half_b<T>(h, co, d.z0 - d.H, d.nzl + 2 * d.H, st->By, st->Bh[0])) return e;
                    st->em_open = true;
                }
                if (int e = em_full_e<T>(h, co, d.z0, d.nzl, st->Bh[st->bh_cur])) return e;
            } else {
                // B half a step on the slab and on the plane below it (the E update of the first owned plane reads it)
                if (int e = em_half_b<T>(h, co, d.z0 - 1, d.nzl + 1)) return e;
                if (int e = em_full_e<T>(h, co, d.z0, d.nzl)) return e;
            }
        } else {
            if (int e = em_half_b<T>(h, co, 0, st->nz)) return e;
            if (int e = em_full_e<T>(h, co, 0, st->nz)) return e;
        }
        timing_end(h);
    }
    if (multi)
        if (int e = exchange<T>(rk, X_EM_E)) return e;
    for (fpic_handle* h : rk.hs) {
        State* st = h->es;
        const Domain& d = *st->dom;
        if (!chain) {
            timing_begin(h, KC_SOLVE);
            const EmCoef<T> co(h);
            if (int e = multi ? em_half_b<T>(h, co, d.z0, d.nzl) : em_half_b<T>(h, co, 0, st->nz)) return e;
            timing_end(h);
        }
        st->substeps_since_bin++;
        st->dom->substeps_since_migration++;
        h->step_launches++;
        h->solve_launches++;
        h->particle_updates += total_particles(st);
    }
    if (multi && !chain)
        if (int e = exchange<T>(rk, X_EM_B)) return e;
    return FPIC_OK;
}

template <typename T>
int dom_substep(Ranks& rk)
{
    if (rk.hs[0]->es->solver == FPIC_SOLVER_YEE) return dom_em_substep<T>(rk);
    State* s0 = rk.hs[0]->es;
    bool unbinned = false;
    for (fpic_handle* h : rk.hs)
        for (const Species& s : h->es->sp) unbinned |= !s.binned;
    if (s0->dom->world > 1) {
        if (unbinned || s0->dom->substeps_since_migration >= s0->dom->migrate_every)
            if (int e = migrate<T>(rk)) return e;
    } else if (unbinned || s0->substeps_since_bin >= 8) {
        for (fpic_handle* h : rk.hs)
            if (int e = bin_all<T>(h, false)) return e;
    }
    // The tile layers along the slab's faces are pushed first; their deposits complete the ghost planes, which then
    // travel (RCCL: on the communicator's stream) while the interior of the slab is pushed.  Every rank takes the same
    // branch: the condition depends on the decomposition and on "every species is binned", which holds on every rank
    // once the first migration has run.
    bool split = true;
    for (fpic_handle* h : rk.hs) split &= can_split(h->es);
    auto count = [](fpic_handle* h) {
        State* st = h->es;
        st->substeps_since_bin++;
        st->dom->substeps_since_migration++;
        h->step_launches++;
        h->particle_updates += total_particles(st);
    };
    if (!split) {
        for (fpic_handle* h : rk.hs) {
            if (int e = deposit_cycle<T, false>(h)) return e;
            count(h);
        }
        return dom_fields<T>(rk, false);
    }
    for (fpic_handle* h : rk.hs)
        if (int e = deposit_cycle<T, false>(h, 1)) return e;
    if (int e = comm_fork(rk)) return e;
    if (int e = exchange<T>(rk, X_GHOST, /*on_comm_stream=*/true)) return e;
    for (fpic_handle* h : rk.hs) {
        if (int e = deposit_cycle<T, false>(h, 2)) return e;
        count(h);
    }
    if (int e = comm_join(rk)) return e;
    return dom_fields<T>(rk, true);
}

int dom_ranks_of(fpic_handle* h, Ranks& rk)
{
    Domain& d = *h->es->dom;
    rk.hs.assign(1, h);
    rk.rccl = h->comm != nullptr;
    if (rk.rccl) {
        if (h->comm->world != d.world || h->comm->rank != d.rank)
            return fail(h, FPIC_ERR_STATE, "the communicator (rank %d of %d) and the decomposition (rank %d of %d) disagree", h->comm->rank, h->comm->world, d.rank, d.world);
    } else if (d.world > 1) {
        return fail(h, FPIC_ERR_STATE, "a decomposed handle steps through its communicator (fpic_comm_init) or its in-process group (fpic_group_step)");
    }
    return FPIC_OK;
}

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
