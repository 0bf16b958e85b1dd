// fpic_handle.hpp — the handle behind the C ABI and the helpers every translation unit of
// libfusionpic.so shares (not part of the ABI).  One handle = one pusher object on one GPU
// (empic.js:30-1529); spec.geometry selects which state hangs off it: the reference's
// axisymmetric (r,z) pusher (fpic_api.hip) or the CART3D electrostatic extension (fes_api.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "fpic_internal.hpp"

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

namespace fpic {
struct BlockWork;
enum KernelClass { KC_PUSH = 0, KC_DEPOSIT, KC_STAMP, KC_PRECALC, KC_SORT, KC_SOLVE, KC_COUNT };
struct PendingTiming {
    hipEvent_t a, b;
    int cls;
};
} // namespace fpic
namespace fes { struct State; }
namespace fcomm { struct Comm; }

struct fpic_handle {
    fpic_spec spec{};
    fpic::Constants k{};
    int prec = FPIC_F32;
    int device = 0;
    int cus = 256;            // compute units of the device: the grid of a persistent launch
    size_t n = 0, n_pad = 0;
    int nr = 0, nz = 0;
    size_t ncell = 0;
    size_t esize = 4;

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // particle state, two sets (binning is out of place), cur selects the live one;
    // the ten arrays of a set are consecutive pieces of one slab, n_pad elements each
    void* slab[2] = {};
    void* part[2][10] = {};
    uint8_t* alive[2] = {};
    uint32_t* id[2] = {};
    int cur = 0;

    // grid state
    void* E = nullptr;
    void* B = nullptr;
    void* sink = nullptr;
    uint8_t* sink_alive = nullptr;
    void* inv_cdf_xy = nullptr;
    void* entropy = nullptr;
    void* coef = nullptr;
    void* cell_sums = nullptr;
    void* moments = nullptr;
    void* norm = nullptr;
    void* avg = nullptr;
    float* stamp = nullptr;
    void* shape_half = nullptr;
    void* shape_tenth = nullptr;
    bool shapes_ready = false;

    // binning by cell tile
    int ntx = 0, ntz = 0;
    uint32_t ntiles = 0; // real tiles + 1 bin for clipped particles
    uint32_t* tile_count = nullptr;
    // two bin tables: [wl] describes the live particle order, [wl ^ 1] is laid out by the
    // next binning (which may be the next push, see scatter_pending)
    uint32_t* tile_start2[2] = {};
    uint32_t* nwork2[2] = {};
    fpic::BlockWork* work2[2] = {};
    int wl = 0;
    uint32_t* tile_cursor = nullptr;
    size_t two_level_min = size_t(1) << 20; // populations from this size on are binned in two levels (FPIC_TWO_LEVEL_MIN overrides: tests)
    uint32_t* coarse_cursor = nullptr; // two-level first binning: first chunk of each coarse bin (sort_chunks_kernel)
    // per work item the neighbour-slot counts of the positions its particles were left at by the last IN-PLACE fused push:
    // what the re-binning launch that follows (same work list, same slots) needs to reserve its ranges without counting
    uint32_t* chunk_census = nullptr;
    bool chunk_census_fresh = false;
    size_t work_cap = 0;
    bool binned = false;
    int deposits_since_bin = 0;
    unsigned long long t_substep = 0; // counter-based RNG mode: global index of the next sub-step
    bool sums_fresh = false;      // cell_sums already holds the sums of the current particle state (fused push)
    bool census_fresh = false;    // tile_count holds the census of the current particle state (fused push)
    bool scatter_pending = false; // tables [wl ^ 1] are laid out from that census: the next push re-bins
    // Particles that missed their LDS tile in a scatter, read back with a lag of two
    // scatters so that the host may run ahead of the GPU by at most two frames.
    unsigned long long* spilled = nullptr;      // device counter
    unsigned long long* spilled_host = nullptr; // pinned, kSpillSlots entries
    hipEvent_t spill_event[2] = {};
    // fpic_density_finish_from on a caller's stream: the grids it writes (moments, norm, avg) are
    // ordered against this handle's own stream through this event
    hipEvent_t finish_event = nullptr;
    hipEvent_t order_event = nullptr; // "everything queued on the handle's stream so far", for a caller's stream to wait on
    bool finish_pending = false;
    bool spill_pending[2] = {};
    unsigned long long deposit_seq = 0;
    unsigned long long last_spill = 0;

    // statistics
    bool profiling = false;
    std::vector<fpic::PendingTiming> pending;
    std::vector<hipEvent_t> event_pool;
    double ms[fpic::KC_COUNT] = {};
    uint64_t particle_updates = 0, step_launches = 0, deposit_launches = 0, sort_passes = 0, solve_launches = 0;
    uint64_t bytes_particles = 0, bytes_grid = 0;

    fes::State* es = nullptr;     // spec.geometry == FPIC_GEOM_CART3D: the electrostatic box (fes_api.hip)
    fcomm::Comm* comm = nullptr;  // fpic_comm_init: RCCL communicator of a multi-GPU run (fpic_comm.cpp)
};

namespace fpic {

std::string& create_error(); // text of the last failed fpic_create on this thread

inline int fail(fpic_handle* h, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else create_error() = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                                      \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fpic::fail(h, e_ == hipErrorOutOfMemory ? FPIC_ERR_OOM : FPIC_ERR_HIP, "%s failed: %s", #expr, \
                              hipGetErrorString(e_));                                                         \
    } while (0)

#define CHECK_HANDLE(h)                                                                                       \
    do {                                                                                                      \
        if (!(h)) return fpic::fail(nullptr, FPIC_ERR_INVALID_ARG, "null handle");                            \
        hipError_t e_ = hipSetDevice((h)->device);                                                            \
        if (e_ != hipSuccess) return fpic::fail(h, FPIC_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e_)); \
    } while (0)

inline unsigned blocks_for(size_t n, unsigned per = 256) { return static_cast<unsigned>((n + per - 1) / per); }

inline int dev_alloc(fpic_handle* h, void** p, size_t bytes, uint64_t* account)
{
    HIP_TRY(h, hipMalloc(p, bytes ? bytes : 16));
    HIP_TRY(h, hipMemsetAsync(*p, 0, bytes ? bytes : 16, h->stream));
    if (account) *account += bytes;
    return FPIC_OK;
}

// HIP-event timing of one kernel class on the handle's stream (only while profiling)
inline void timing_begin(fpic_handle* h, int cls)
{
    if (!h->profiling) return;
    PendingTiming t;
    t.cls = cls;
    for (hipEvent_t* e : { &t.a, &t.b }) {
        if (!h->event_pool.empty()) { *e = h->event_pool.back(); h->event_pool.pop_back(); }
        else (void)hipEventCreate(e);
    }
    (void)hipEventRecord(t.a, h->stream);
    h->pending.push_back(t);
}

inline void timing_collect(fpic_handle* h)
{
    if (h->pending.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (PendingTiming& t : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) h->ms[t.cls] += ms;
        h->event_pool.push_back(t.a);
        h->event_pool.push_back(t.b);
    }
    h->pending.clear();
}

inline void timing_end(fpic_handle* h)
{
    if (!h->profiling) return;
    (void)hipEventRecord(h->pending.back().b, h->stream);
    if (h->pending.size() > 2048) timing_collect(h);
}

} // namespace fpic
