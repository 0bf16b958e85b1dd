// fes_host_push.inc.hpp: allocation of a species, the launches of the tiled push (one launch for every binned species) and the binning passes — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.

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
    const bool em = st->solver == FPIC_SOLVER_YEE; // (the full-EM tile shape: EmWin)
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
    const bool em = st->solver == FPIC_SOLVER_YEE; // (the full-EM tile shape: EmWin)
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
