// fes_host_solve.inc.hpp: the Poisson solve of one handle (the library's own transform passes or rocFFT), the two-part push's layer rule, the electrostatic sub-step — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.
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
            em_edge_gradient_kernel<T><<<node_launch(st->nx, st->ny, st->nz).grid, node_launch(st->nx, st->ny, st->nz).block, 0, h->stream>>>(static_cast<const T*>(st->phi), st->nx, st->ny, st->nz,
                                                                                   static_cast<T>(1.0 / (st->lx / st->nx)), static_cast<T>(1.0 / (st->ly / st->ny)),
                                                                                   static_cast<T>(1.0 / (st->lz / st->nz)), static_cast<T*>(st->Ey), 0, st->nz, held_of(st));
            HIP_TRY(h, hipGetLastError());
        } else {
            gradient_kernel<T><<<node_launch(st->nx, st->ny, st->nz).grid, node_launch(st->nx, st->ny, st->nz).block, 0, h->stream>>>(
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
            em_edge_gradient_kernel<T><<<node_launch(st->nx, st->ny, st->nz).grid, node_launch(st->nx, st->ny, st->nz).block, 0, h->stream>>>(static_cast<const T*>(st->phi), st->nx, st->ny, st->nz,
                                                                                   static_cast<T>(1.0 / (st->lx / st->nx)), static_cast<T>(1.0 / (st->ly / st->ny)),
                                                                                   static_cast<T>(1.0 / (st->lz / st->nz)), static_cast<T*>(st->Ey), 0, st->nz, held_of(st));
        else
            gradient_kernel<T><<<node_launch(st->nx, st->ny, st->nz).grid, node_launch(st->nx, st->ny, st->nz).block, 0, h->stream>>>(
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
