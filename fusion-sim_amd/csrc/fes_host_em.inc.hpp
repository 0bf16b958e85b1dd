// fes_host_em.inc.hpp: the full-EM cycle of one handle (node centring, push + current, the lattice sweeps, the chained lattice step) — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.
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
    if (nk <= 0) return FPIC_OK;
    const NodeLaunch nl = node_launch(st->nx, st->ny, nk);
    em_update_e_kernel<T><<<nl.grid, nl.block, 0, h->stream>>>(static_cast<T*>(st->Ey), static_cast<const T*>(b ? b : st->By), st->Jfix, st->nx,
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
