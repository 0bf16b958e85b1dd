// fpic_comm.cpp — fpic_comm_* (include/fusionpic.h): RCCL communicator of a handle.
//
// Reference-parity mode: particles are sharded by index range, the grid
// tables are replicated and the only exchange per frame is ONE all-reduce (sum) of the per-cell sums
// between the scatter and the stamp / normalise / EMA stage of density() — issued by the library
// itself, so a JavaScript host can shard a run without any other collective library.  Because the
// deposit is never fed back into the push (empic.js:1471-1505) the exchange is taken off the
// critical path: the sums are copied to a buffer of their own, the all-reduce and the finish stage
// run on a side stream, and the next step() starts at once on the handle's stream.
#include "fpic_comm.hpp"

#include <cstring>
#include <new>

using namespace fpic;

namespace fcomm {

int check(fpic_handle* h, ncclResult_t r, const char* what)
{
    if (r == ncclSuccess) return FPIC_OK;
    const fdyn::Rccl& rc = fdyn::rccl();
    return fail(h, FPIC_ERR_HIP, "%s failed: %s", what, rc.ok && rc.GetErrorString ? rc.GetErrorString(r) : "RCCL error");
}

void release(fpic_handle* h)
{
    Comm* c = h->comm;
    if (!c) return;
    if (c->side) (void)hipStreamSynchronize(c->side);
    const fdyn::Rccl& rc = fdyn::rccl();
    if (c->nccl && rc.ok) (void)rc.CommDestroy(c->nccl);
    if (c->buf) (void)hipFree(c->buf);
    if (c->copied) (void)hipEventDestroy(c->copied);
    if (c->reduced) (void)hipEventDestroy(c->reduced);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    h->comm = nullptr;
}

} // namespace fcomm

extern "C" {

int fpic_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".id <- Non-optional property is undefined!");
    const fdyn::Rccl& rc = fdyn::rccl();
    if (!rc.ok) return fail(nullptr, FPIC_ERR_STATE, "RCCL is not available (%s)", rc.why.c_str());
    ncclUniqueId id;
    static_assert(sizeof id == FPIC_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = rc.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, FPIC_ERR_HIP, "ncclGetUniqueId failed: %s", rc.GetErrorString(r));
    std::memcpy(id128, &id, sizeof id);
    return FPIC_OK;
}

int fpic_comm_init(fpic_handle* h, const void* id128, int rank, int world)
{
    CHECK_HANDLE(h);
    if (!id128) return fail(h, FPIC_ERR_INVALID_ARG, ".id <- Non-optional property is undefined!");
    if (world < 1 || rank < 0 || rank >= world) return fail(h, FPIC_ERR_INVALID_ARG, ".rank <- %d is outside a world of %d", rank, world);
    if (h->comm) return fail(h, FPIC_ERR_STATE, "the handle already has a communicator");
    const fdyn::Rccl& rc = fdyn::rccl();
    if (!rc.ok) return fail(h, FPIC_ERR_STATE, "RCCL is not available (%s)", rc.why.c_str());
    fcomm::Comm* c = new (std::nothrow) fcomm::Comm();
    if (!c) return fail(h, FPIC_ERR_OOM, "host allocation failed");
    h->comm = c;
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    auto bail = [&](int code) { fcomm::release(h); return code; };
    if (int e = fcomm::check(h, rc.CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank")) return bail(e);
    // the exchange overlaps the next frame's push: its few workgroups must be placed as soon as a push
    // workgroup retires, hence the high priority
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e;
    if ((e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->copied, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->reduced, hipEventDisableTiming)) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "communicator setup failed: %s", hipGetErrorString(e)));
    return FPIC_OK;
}

int fpic_comm_destroy(fpic_handle* h)
{
    CHECK_HANDLE(h);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    fcomm::release(h);
    return FPIC_OK;
}

int fpic_comm_info(fpic_handle* h, int* rank, int* world)
{
    CHECK_HANDLE(h);
    if (rank) *rank = h->comm ? h->comm->rank : 0;
    if (world) *world = h->comm ? h->comm->world : 1;
    return FPIC_OK;
}

int fpic_comm_set_overlap(fpic_handle* h, int enable)
{
    CHECK_HANDLE(h);
    if (!h->comm) return fail(h, FPIC_ERR_STATE, "the handle has no communicator (fpic_comm_init)");
    h->comm->overlap = enable != 0;
    return FPIC_OK;
}

} // extern "C"
