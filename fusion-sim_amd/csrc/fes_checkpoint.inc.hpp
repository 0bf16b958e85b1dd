// fes_checkpoint.inc.hpp: checkpoint files of an undecomposed box and of one rank of a decomposition — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.
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
