// fes_state.inc.hpp: the state of a CART3D handle (Species, State) and of its z-slab decomposition (Domain) — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.

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
    int ltx = FES_LTX, lty = FES_LTY, ltz = FES_LTZ; // log2 of the tile edges: 16x16x8 cells (electrostatic), 8x8x8 (full EM)
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
