/*
 * fusionpic.h — C ABI of libfusionpic.so, the MI355X (gfx950) implementation of
 * fusion-sim's per-step particle-in-cell hot path.
 *
 * The library replaces the object returned by the reference factory
 *     empic.makeCylindricalParticlePusher(spec)          (empic.js:30-1529)
 * for the calls that sit on the hot path.  Every entry point names the
 * reference interface it stands in for (file:line under
 * /root/reference/public/javascripts/).  The JavaScript host keeps the reference's
 * method names through an N-API addon (fusion-sim_amd/js/); the binding a
 * maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  No C++ exception crosses the ABI.
 *   - Every call returns FPIC_OK (0) or a negative fpic_status.  The text of the
 *     last failure on a handle is fpic_last_error(h); for a failed fpic_create it
 *     is fpic_last_error(NULL).  The N-API layer turns a non-zero status into a
 *     synchronous `throw new Error(msg)`, which is how the reference reports
 *     spec and GL failures (utilities.js:118-127, :213-259).
 *   - Host buffers are caller-owned and copied during the call; no host pointer
 *     is kept.  Device buffers belong to the handle and die with fpic_destroy.
 *   - A handle is not thread-safe (the reference runs on one JS thread).
 *   - Calls enqueue on the handle's HIP stream and return; fpic_read_*,
 *     fpic_get_particles and fpic_sync wait for the stream.
 *   - There is NO CPU fallback: without a gfx950 device fpic_create fails with
 *     FPIC_ERR_NO_DEVICE.
 *
 * Grid layout at the boundary (reference: empic.js:1162, texture index
 * 4*(i + j*nr) + c with i = r index, j = z index): fpic_read_grid returns
 * exactly that RGBA float layout.  fpic_set_grid takes the reference's nested
 * JavaScript order value[i][j][k], flattened as ((i*nz + j)*ncomp + k).
 */
#ifndef FUSIONPIC_H
#define FUSIONPIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPIC_ABI_VERSION 2

typedef enum fpic_status {
    FPIC_OK = 0,
    FPIC_ERR_INVALID_ARG = -1, /* bad spec / argument; message keeps ".prop <- ..." form */
    FPIC_ERR_NO_DEVICE = -2,   /* no gfx950 device visible: the product path has no CPU fallback */
    FPIC_ERR_HIP = -3,         /* a HIP runtime call failed */
    FPIC_ERR_OOM = -4,         /* device or host allocation failed */
    FPIC_ERR_STATE = -5        /* call made in the wrong state (e.g. step before set) */
} fpic_status;

typedef enum fpic_dtype {
    FPIC_F32 = 0,
    FPIC_F64 = 1
} fpic_dtype;

/* Grids accepted by fpic_set_grid (reference: out.set, empic.js:1157-1350). */
typedef enum fpic_grid_in {
    FPIC_GRID_E = 0,          /* [nr][nz][3]  V/m   (empic.js:1159-1177) */
    FPIC_GRID_B = 1,          /* [nr][nz][3]  T     (empic.js:1179-1197) */
    FPIC_GRID_SINK_MASK = 2,  /* [nr][nz]     alive where > 0.5 (empic.js:1246-1260) */
    FPIC_GRID_SOURCE_PDF = 3  /* [nr][nz]     un-normalised pdf -> 512x512 inverse CDF (empic.js:1263-1349) */
} fpic_grid_in;

/* Grids returned by fpic_read_grid, all RGBA, index 4*(i + j*W) + c. */
typedef enum fpic_grid_out {
    FPIC_READ_MOMENTS = 0, /* moments01      nr x nz  (empic.js:933, K4)  */
    FPIC_READ_NORM = 1,    /* moments01_norm nr x nz  (empic.js:1040, K5) */
    FPIC_READ_AVG = 2,     /* moments01_avgA nr x nz  (empic.js:1071, K6) */
    FPIC_READ_R1 = 3,      /* nr x nz, w = 1 (empic.js:506-542) */
    FPIC_READ_R2 = 4,      /* (empic.js:545-581) */
    FPIC_READ_R3 = 5,      /* (empic.js:585-621) */
    FPIC_READ_A = 6,       /* (empic.js:625-659) */
    FPIC_READ_B = 7,       /* B frame buffer incl. painters' alpha (empic.js:197) */
    FPIC_READ_E = 8,       /* (empic.js:186) */
    FPIC_READ_SINK = 9,    /* red channel only is written (empic.js:1249) */
    FPIC_READ_INV_CDF = 10 /* 512 x 512, .xy used (empic.js:228-241, :1328-1339) */
} fpic_grid_out;

/* ---- extension: spec.geometry.  The reference has ONE geometry (axisymmetric (r,z) grid, static
 * fields, empic.js:111-115) and no field solve in its step loop (empic.js:1436-1505; SURVEY.md
 * section 0).  FPIC_GEOM_CART3D is the self-consistent electrostatic mode of BASELINE.json
 * configs[2..4]: a periodic box radius x length_y x height (x, y, z) on nr x ny x nz nodes,
 * CIC gather/deposit, Poisson solve each sub-step.  It has no reference counterpart (parity
 * unpinned; defined by oracle/es3d_oracle_impl.h).  The same entry points drive it:
 *   fpic_set_particles  positions in metres (wrapped into the box), velocities in units of c
 *   fpic_add_bz / fpic_add_b   uniform external magnetic field (Boris rotation)
 *   fpic_precalc        fields <- particles: deposit + solve (the fields->coefficients stage,
 *                       empic.js:1413-1434; must precede the first step)
 *   fpic_step           2 sub-steps per call (empic.js:1436-1469), each push+deposit fused, then solve
 *   fpic_density        no-op: the charge density of the current positions is always at hand
 *   fpic_get_particles / fpic_get_cells / fpic_sort / fpic_sync / fpic_profile / fpic_get_stats
 * Calls that only make sense on the (r,z) pusher return FPIC_ERR_STATE. */
typedef enum fpic_geometry {
    FPIC_GEOM_CYL_RZ = 0, /* the reference's pusher */
    FPIC_GEOM_CART3D = 1
} fpic_geometry;

typedef enum fpic_solver {
    FPIC_SOLVER_NONE = 0,        /* fields are what fpic_set_field3 uploaded */
    FPIC_SOLVER_POISSON_FFT = 1, /* rocFFT forward/inverse around a hand-written k-space kernel */
    FPIC_SOLVER_YEE = 2          /* full EM (BASELINE configs[4]): Yee FDTD, node-centred CIC gather of E and B, charge-conserving
                                    integer current deposit (zigzag); precalc() sets E to the Poisson field of the charge on the
                                    lattice's edges and B to the uniform external field; density() deposits the charge grid */
} fpic_solver;

/* spec.shape: the deposit of density() on the (r,z) grid (SURVEY.md 8(b) key shape:'ref11'|'cic'). */
typedef enum fpic_shape {
    FPIC_SHAPE_REF11 = 0, /* the reference's 11x11 cos^2 point sprite (empic.js:949-1035) */
    FPIC_SHAPE_CIC = 1    /* extension, no reference counterpart: bilinear over the four nearest cell centres */
} fpic_shape;

/* CART3D grids, node index i + nr*(j + ny*k) (i fastest, as the reference's texel index
 * 4*(i + j*nr), empic.js:1162). */
typedef enum fpic_field3 {
    FPIC_F3_E = 0,         /* in: value[i][j][k][3] V/m;  out: 4 per node (Ex, Ey, Ez, phi) */
    FPIC_F3_RHO = 1,       /* out: C/m^3, 1 per node */
    FPIC_F3_PHI = 2,       /* out: V, 1 per node */
    FPIC_F3_RHO_FIXED = 3, /* out: int64 per node, 2^42 per unit charge number (exact, order-free) */
    /* full EM only; lattice arrays are 4 per node: E = (Ex(i+1/2,j,k), Ey(i,j+1/2,k), Ez(i,j,k+1/2), 0), B on the faces */
    FPIC_F3_B_NODES = 4,   /* out: node-centred B, 4 per node */
    FPIC_F3_EDGE_E = 5,    /* in: value[i][j][k][3]; out: 4 per node */
    FPIC_F3_FACE_B = 6,    /* in: value[i][j][k][3]; out: 4 per node */
    FPIC_F3_J_FIXED = 7    /* out: 3 int64 per node (x-, y-, z-edge), 96 * 2^42 per particle crossing a dual face: with RHO_FIXED the
                              lattice continuity equation 96 (rho^(n+1) - rho^n) + div J = 0 holds exactly */
} fpic_field3;

/* Device buffers whose address can be handed to a collective (see fpic_device_buffer). */
typedef enum fpic_buffer {
    FPIC_BUF_CELL_SUMS = 0, /* per-cell sums 0.001*(vr,vtheta,vz,1), (nr+11)*(nz+11)*4 scalars: sprite-centre cell (ic, jc),
                               ic in -5..nr+5, at 4*((ic+5) + (nr+11)*(jc+5)); opaque to a host that only sums it over ranks */
    FPIC_BUF_RHO_FIXED = 1  /* CART3D: int64 charge accumulators, nr*ny*nz */
} fpic_buffer;

/*
 * Construction parameters.  The first eight fields are the reference's spec
 * (empic.js:31-41, all 'number').  `nparticles` is the side of the reference's
 * particle texture: the particle count is nparticles*nparticles (empic.js:107-109)
 * unless `count` is non-zero (extension).
 */
typedef struct fpic_spec {
    double radius;          /* m */
    double height;          /* m */
    int32_t nr;
    int32_t nz;
    double dt;              /* s, fixed for the life of the handle (empic.js:44, :852) */
    int32_t nparticles;     /* texture side; count = nparticles^2 */
    double particle_mass;   /* kg */
    double particle_charge; /* C  */
    /* ---- extensions (zero = reference behaviour) ---- */
    uint64_t count;         /* exact particle count overriding nparticles^2 */
    int32_t precision;      /* fpic_dtype of the device state; reference is F32 */
    int32_t device;         /* HIP device ordinal */
    int32_t physical_a;     /* 0: reference's K9 formula incl. quirk Q1 (empic.js:645);
                               1: h(E.B)B vector form */
    int32_t sort_interval;  /* re-bin particles every k density() calls; 0 = adaptive */
    int32_t unfused_deposit;/* 0: step() also forms the per-cell sums of density()'s scatter, counts
                               particles per tile and re-bins them (the frame loop of
                               fusionsim.js:172-174 always calls density() after step());
                               1: separate passes for all of that;
                               2: census and re-binning stay in step(), the sums are a separate pass
                               (faster with rng_mode 1, where no gather hides the LDS atomics) */
    int32_t rng_mode;       /* 0: the reference's generator (entropy-table walk K3, per-particle state,
                               empic.js:783-820).  1: counter-based extension: the random
                               vector of particle i at sub-step t is Philox4x32-10(counter (i, t, 0x5EED),
                               key rng_seed); no per-particle random state, no entropy table */
    uint32_t rng_seed_lo, rng_seed_hi;
    int32_t geometry;       /* fpic_geometry; 0 = the reference's (r,z) pusher */
    int32_t solver;         /* fpic_solver (CART3D) */
    int32_t ny;             /* CART3D: nodes along y (nr along x, nz along z) */
    int32_t shape;          /* fpic_shape of density() on the (r,z) grid; CART3D always deposits CIC */
    double length_y;        /* CART3D: box is radius (x) x length_y x height (z) metres */
    double macro_weight;    /* CART3D: real particles per macro-particle (charge density scale); 0 = 1 */
    int32_t raster_subpixel_bits; /* density()'s point sprites (empic.js:980-1035, :1473-1478) on the (r,z) grid.
                               0: ideal sprites — window coordinates of infinite precision, a point whose centre lies
                               outside the clip volume is discarded whole (the GL ES 2.0 text).
                               b = 1..8: as a rasteriser with b sub-pixel bits draws them — window position snapped to
                               2^-b pixel (round half to even, y running downwards), left/top edges inclusive, a point whose
                               centre has left the target cropped instead of discarded.  b = 4 reproduces the reference
                               run under Chromium's ANGLE/SwiftShader bit for bit (tests/golden/webgl_*); desktop GPUs
                               usually have 8.  Only the cell a particle's 11x11 stamp is centred on changes (one cell
                               lower for a coordinate within 2^-(b+1) pixel above a pixel edge).  Carved out of the
                               former reserved[6]: same size and offsets, ABI unchanged */
    int32_t reserved_i32;
    double reserved[5];
} fpic_spec;

typedef struct fpic_handle fpic_handle;

/* Counters and timings; all times are HIP-event milliseconds on the handle's stream
 * and are gathered only while profiling is enabled (fpic_profile). */
typedef struct fpic_stats {
    uint64_t n_particles;
    uint64_t particle_updates;   /* sub-steps x particles since create */
    uint64_t step_launches;      /* push kernel launches */
    uint64_t deposit_launches;
    uint64_t sort_passes;
    uint64_t deposit_spilled;    /* particles of the last deposit that missed their LDS tile */
    double ms_push;              /* sum over push launches */
    double ms_deposit;           /* sum over cell-sum (scatter) launches */
    double ms_stamp;             /* sum over stamp-convolution + normalise + EMA launches */
    double ms_precalc;
    double ms_sort;
    uint64_t bytes_particle_state; /* device bytes held for particle state */
    uint64_t bytes_grid_state;
    double ms_solve;             /* CART3D: sum over field solves (rho conversion, FFTs, k-space, gradient) */
    uint64_t solve_launches;
    double reserved[6];
} fpic_stats;

/* Last error text.  h may be NULL after a failed fpic_create. */
const char* fpic_last_error(const fpic_handle* h);

/* Library/ABI version, and the code-object architecture it was built for ("gfx950"). */
int fpic_abi_version(void);
const char* fpic_build_arch(void);

/* empic.makeCylindricalParticlePusher(spec): validation (empic.js:31-41,
 * utilities.js:118-127), derived constants h, factor_r, factor_z (empic.js:44-46),
 * all state buffers (empic.js:123-241, :499-502, :666-672, :933-1072), the 11x11
 * stamp (empic.js:949-971). */
int fpic_create(const fpic_spec* spec, fpic_handle** out);
int fpic_destroy(fpic_handle* h);

/* out.set({position, velocity}) (empic.js:1199-1244).  pos/vel are AoS [n][3] in
 * metres / units of c, dtype F32 or F64; either may be NULL.  n must equal the
 * handle's particle count.  Normalisation by factor_r, factor_r, factor_z is done
 * in double and rounded once, as the reference's Float32Array store does. */
int fpic_set_particles(fpic_handle* h, const void* pos_aos, const void* vel_aos, uint64_t n, int dtype);

/* out.set({E, B, sink_mask, source_pdf}) (empic.js:1159-1197, :1246-1349).
 * data is value[i][j][k] flattened, i over nr, j over nz, k over ncomp (3 or 1). */
int fpic_set_grid(fpic_handle* h, int which, const void* data, int nr, int nz, int ncomp, int dtype);

/* Reproducible replacement for window.crypto / Math.random (empic.js:142-180, quirk
 * Q8): entropy is 1024*1024*4 floats (index 4*(i + 1024*j)), rand is n*4 floats
 * (u1,u2,c1,c2).  Either may be NULL to keep the current one. */
int fpic_set_random_state(fpic_handle* h, const float* entropy, const float* rand);

/* Static field painters, additive into B (empic.js:1352-1363, :1380-1411). */
int fpic_add_current_loop(fpic_handle* h, double r, double z, double current);
int fpic_add_current_z(fpic_handle* h, double current);
int fpic_add_bz(fpic_handle* h, double bz);
int fpic_add_btheta(fpic_handle* h, double btheta);

/* out.precalc() (empic.js:1413-1434): B,E -> R1,R2,R3,A. */
int fpic_precalc(fpic_handle* h);

/* out.step() (empic.js:1436-1469) ncalls times; each call is two leap-frog
 * sub-steps (RandB,VelB,PosB,RandA,VelA,PosA). */
int fpic_step(fpic_handle* h, int ncalls);
/* The same advance counted in single leap-frog sub-steps: fpic_step(h, n) == fpic_substeps(h, 2 n).  The reference can
 * only advance in pairs (empic.js:1436-1469 draws both halves of the ping-pong); an odd count exists for diagnostics that
 * need the state between the two halves (the continuity check of the full-EM cycle, a sampled comparison with the oracle
 * in a field read back beforehand). */
int fpic_substeps(fpic_handle* h, int nsub);

/* out.density() (empic.js:1471-1495): scatter (K4), normalise (K5), EMA (K6),
 * avgB <- avgA (K7).  fpic_density == fpic_deposit then fpic_density_finish; the
 * split exists so that a multi-GPU host can sum FPIC_BUF_CELL_SUMS across ranks
 * in between. */
int fpic_density(fpic_handle* h);
int fpic_deposit(fpic_handle* h);
int fpic_density_finish(fpic_handle* h);
/* fpic_density_finish reading the per-cell sums from a caller's device buffer (same layout as
 * FPIC_BUF_CELL_SUMS) and running on a caller's stream (NULL: the handle's).  A multi-GPU host
 * copies the sums out after fpic_deposit, all-reduces the copy and finishes from it on a side
 * stream while the next step() already runs; the library orders its own later reads of the
 * density grids after that finish. */
int fpic_density_finish_from(fpic_handle* h, const void* sums, void* hip_stream);

/* fb.readPixels (utilities.js:701-711) for the grids above.  out holds
 * 4*W*H floats (dtype F32) or doubles (F64). */
int fpic_read_grid(fpic_handle* h, int which, void* out, int dtype);

/* Particle read-back in the caller's original order (normalised units as stored:
 * x/R, y/R, z/H and v/c scaled the same way).  pos_aos/vel_aos are [n][3] of
 * dtype, rand is [n][4] float, alive is [n] bytes; any may be NULL. */
int fpic_get_particles(fpic_handle* h, void* pos_aos, void* vel_aos, float* rand, uint8_t* alive, int dtype);

/* NGP cell index i + j*nr of every particle as the push sees it (integer parity
 * check; same order as fpic_get_particles). */
int fpic_get_cells(fpic_handle* h, int32_t* cells);

/* Multi-GPU plumbing: the raw device address/byte size of a buffer (for an RCCL
 * collective issued by the host), and the HIP stream the handle enqueues on.
 * fpic_set_stream(h, NULL) restores the handle's own stream. */
int fpic_device_buffer(fpic_handle* h, int which, void** dptr, size_t* bytes);
int fpic_set_stream(fpic_handle* h, void* hip_stream);
int fpic_get_stream(fpic_handle* h, void** hip_stream);

/* ---- CART3D extension entry points (FPIC_ERR_STATE on an (r,z) handle) ---- */
/* A further species sharing the grid (the reference has one species per pusher, empic.js:37-38).
 * charge must be a non-zero integer multiple (|Z| <= 255) of spec.particle_charge; the macro weight
 * is shared.  *index receives the species number (spec's own species is 0). */
int fpic_add_species(fpic_handle* h, double mass, double charge, uint64_t count, int* index);
int fpic_set_particles_of(fpic_handle* h, int species, const void* pos_aos, const void* vel_aos, uint64_t n, int dtype);
/* the caller's particles [first, first + n) of a species: populations too large for one host array are
 * uploaded piecewise (2e9 particles are 48 GB of AoS floats).  pos_aos / vel_aos here and in
 * fpic_domain_set_particles may be host OR device memory (copied with hipMemcpyDefault on the handle's stream;
 * for device memory the caller has synchronised whatever produced it). */
int fpic_set_particles_range(fpic_handle* h, int species, uint64_t first, uint64_t n, const void* pos_aos, const void* vel_aos, int dtype);
int fpic_get_particles_of(fpic_handle* h, int species, void* pos_aos, void* vel_aos, int dtype);
int fpic_get_cells_of(fpic_handle* h, int species, int32_t* cells);
/* The mirror of fpic_set_particles_range, and a sampled read-back: the caller's particles first, first + stride,
 * first + 2 stride, ... — n of them (stride >= 1; stride 1 = the contiguous range [first, first + n)) — in the caller's
 * order, pos_aos / vel_aos [n][3] of dtype (either may be NULL).  The reference can only display its particles
 * (fusionsim.js:174-178; utilities.js:701-711 has readPixels, unused): 2e9 particles are 48 / 96 GB of host arrays, so a
 * host reads them piecewise or samples them (every 20 000th particle of BASELINE configs[4] for the oracle comparison). */
int fpic_get_particles_range(fpic_handle* h, int species, uint64_t first, uint64_t n, uint64_t stride, void* pos_aos, void* vel_aos, int dtype);
int fpic_get_cells_range(fpic_handle* h, int species, uint64_t first, uint64_t n, uint64_t stride, int32_t* cells);
/* uniform external B (T), additive like the reference's painters (empic.js:1391-1400) */
int fpic_add_b(fpic_handle* h, double bx, double by, double bz);
/* which = FPIC_F3_E only: value[i][j][k][3] flattened, dims must equal the handle's */
int fpic_set_field3(fpic_handle* h, int which, const void* data, int nx, int ny, int nz, int dtype);
/* out: nodes (RHO, PHI), 4*nodes (E) of dtype, or nodes int64 (RHO_FIXED, dtype ignored) */
int fpic_read_field3(fpic_handle* h, int which, void* out, int dtype);

/* ---- multi-GPU inside the library (the reference has one WebGL context and no
 * communication).  One process per GPU, one handle per process; the library binds RCCL at run time and
 * issues the collectives itself, so the JavaScript host needs no other collective library.
 *   rank 0: fpic_comm_unique_id(id); the host hands the 128 bytes to every rank (file, socket, env);
 *   every rank: fpic_comm_init(h, id, rank, world).
 * (r,z) pusher, reference-parity mode: each rank holds a shard of the particles (contiguous index range)
 * and a replica of the grid tables; fpic_density then sums FPIC_BUF_CELL_SUMS over the ranks with ONE
 * all-reduce between the scatter and the stamp stage — by default on a side stream, off the critical path
 * of the next fpic_step (fpic_comm_set_overlap(h, 0) keeps it on the handle's stream). */
#define FPIC_UNIQUE_ID_BYTES 128
int fpic_comm_unique_id(void* id128);
int fpic_comm_init(fpic_handle* h, const void* id128, int rank, int world);
int fpic_comm_destroy(fpic_handle* h);
int fpic_comm_info(fpic_handle* h, int* rank, int* world);
int fpic_comm_set_overlap(fpic_handle* h, int enable);

/* ---- CART3D spatial decomposition: z-slabs.  spec describes the GLOBAL box and grid on
 * every rank; spec.count is the rank's CAPACITY per species.  Rank r of `world` owns the particles whose cell lies
 * in the planes [r nz/world, (r+1) nz/world); they may sit up to ghost_planes planes outside it between two
 * migrations (every migrate_every sub-steps).  Per sub-step the ranks exchange: the ghost planes of the int64
 * charge grid with their two neighbours (added exactly), then an all-gather of the owned planes of rho; every rank
 * then solves the fields.  At a migration: two counts and two particle messages (6 scalars + the caller's global
 * index) per neighbour, grouped ncclSend/ncclRecv.  Because the charge grid is an integer grid, an N-rank run
 * reproduces the one-GPU run bit for bit.
 *   with a communicator (fpic_comm_init, one process per GPU): fpic_precalc / fpic_step exchange over RCCL;
 *   fpic_group_precalc / fpic_group_step: all `n` ranks are handles of THIS process on one device and the exchange
 *   is device-to-device copies — the stand-in that lets one GPU run and test an N-rank decomposition. */
/* distributed_solve = 0: every rank gathers rho and transforms the whole grid (the N-rank run is then bit-identical to one GPU);
 * 1: the Poisson solve is decomposed too — 2-D transforms of the owned planes, an all-to-all transposition (each pair of ranks
 * exchanges nz/N * ny/N * (nx/2+1) complex values), transforms along z on ny/N rows, the transposition back, and G+1 / G+2 planes
 * of the potential from the neighbours: no rank touches the whole grid, fields agree with one GPU to rounding (needs ny % N == 0).
 * On power-of-two grids (the library's own transforms) the decomposed solve is the one handle's bit for bit, full-EM handles
 * take it for their initial field as well, and the rank then KEEPS ONLY ITS SLAB: nz/N + 2 (ghost_planes + 2) + 1 planes of
 * every node array instead of nz (FPIC_DOMAIN_COMPACT=0 keeps whole-grid arrays).  Call it on a fresh handle: node fields
 * uploaded before it are dropped (upload after).  fpic_read_field3 still fills a whole-grid-shaped array: the planes the
 * rank holds in their places, zero elsewhere; fpic_device_buffer(FPIC_BUF_RHO_FIXED) is then the held planes only.
 * 2: decomposed WITHOUT the transpositions (power-of-two grids, up to 8 ranks): K^2 = k2x + k2y + k2z with k2z the eigenvalues
 * of the three-point second difference, so after the x and y transforms of its own planes every (kx, ky) mode is a periodic
 * tridiagonal system along z; each rank eliminates its nz/N planes to a two-equation interface, ONE all-gather carries two
 * planes of the half spectrum per rank (1/32 of the transpositions' bytes at nz/N = 64), every rank solves the 2N-unknown
 * interface system of each mode redundantly and substitutes back (csrc/fes_tri.hpp).  Equal to the transform solve in exact
 * arithmetic: fields agree with one GPU to rounding (2e-5 / 1e-10), not bit for bit.  Slab-only arrays as with 1. */
int fpic_domain_init(fpic_handle* h, int rank, int world, int ghost_planes, int migrate_every, int distributed_solve);
/* the rank's initial particles (positions anywhere in its slab +- ghost planes); their global indices are first_id, first_id+1, ... */
int fpic_domain_set_particles(fpic_handle* h, int species, uint64_t n, const void* pos_aos, const void* vel_aos, uint32_t first_id, int dtype);
/* the particles the rank holds now, in no particular order, with their global indices; *n receives the count
 * (pass NULL buffers to query it) */
int fpic_domain_get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, uint32_t* ids, uint64_t capacity, uint64_t* n, int dtype);
int fpic_domain_stats(fpic_handle* h, uint64_t* migrated, uint64_t* lost);
int fpic_group_precalc(fpic_handle** handles, int n);
int fpic_group_step(fpic_handle** handles, int n, int ncalls);
/* out.density() (empic.js:1471) of every member of a full-EM group: the charge grid of the current positions, complete on
 * every member's own planes (ghost planes exchanged and added).  Over a communicator every rank calls fpic_density. */
int fpic_group_density(fpic_handle** handles, int n);

/* Counter-based RNG mode only: the global sub-step index (starts at 0, +2 per step() call);
 * settable so that a run can be resumed. */
int fpic_get_substep_counter(fpic_handle* h, uint64_t* t);
int fpic_set_substep_counter(fpic_handle* h, uint64_t t);

/* Force a re-bin of the particle arrays by cell tile now (normally automatic). */
int fpic_sort(fpic_handle* h);

/* Read-back / resume (the reference can only display its state,
 * utilities.js:701-711 is unused): a flat binary dump of the particle arrays in the caller's
 * order and of the grid tables, and its inverse.  The target handle of a load must have been
 * created from the same spec.  A run resumed from a checkpoint continues bit-identically.
 * A box (spec.geometry = CART3D) saves the raw state of every species and its fields (its own file layout); the
 * target handle must have the same species added.  A rank of a decomposition writes / reads its own file (the particles
 * it holds with their global indices; full EM: the lattice fields of its planes; the electrostatic field is recomputed
 * by fpic_precalc after the load). */
int fpic_save_checkpoint(fpic_handle* h, const char* path);
int fpic_load_checkpoint(fpic_handle* h, const char* path);

int fpic_sync(fpic_handle* h);
int fpic_profile(fpic_handle* h, int enable);
int fpic_get_stats(fpic_handle* h, fpic_stats* out);
int fpic_reset_stats(fpic_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* FUSIONPIC_H */
