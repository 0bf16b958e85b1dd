// fes_domain.inc.hpp: the z-slab decomposition — message lists, the two transports (RCCL, in-process group), migration, the decomposed solves, the decomposed cycles — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.

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
        if (count <= 0) return FPIC_OK;
        const NodeLaunch nl = node_launch(st->nx, st->ny, count);
        gradient_planes_kernel<T><<<nl.grid, nl.block, 0, h->stream>>>(
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
        if (st->solver == FPIC_SOLVER_YEE) { // the initial E on the edges of the slab and of its H halo planes on either side
            const int count = d.nzl + 2 * d.H;
            em_edge_gradient_kernel<T><<<node_launch(st->nx, st->ny, count).grid, node_launch(st->nx, st->ny, count).block, 0, h->stream>>>(
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
