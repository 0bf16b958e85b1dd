// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for the ten RCCL entry points libfusionpic.so binds
// (csrc/fpic_dyn.cpp), so that the library's RCCL transport — one handle per rank, grouped ncclSend/ncclRecv of the
// decomposition, ncclAllGather / ncclAllReduce — can be driven by N threads of ONE process on ONE GPU, where the real
// RCCL refuses two ranks on one device.  Bound through FPIC_RCCL_LIBRARY by tests/test_gpu_fake_rccl.py only.
//
// Semantics kept: ranks of a communicator meet by its unique id; between a pair of ranks the k-th send matches the k-th
// receive in issue order; operations of a group take effect at ncclGroupEnd; point-to-point operations involve their
// two ranks only (a rank with nothing to send or receive in a round takes no part in it: no barrier over the world);
// collectives involve every rank.  Simplification: operations complete synchronously (the caller's stream is drained
// first, the copy is a blocking device-to-device copy), which is stricter than stream-ordered execution; where the real
// library would wait for ever — a receive whose send never comes, a send nobody receives — the stand-in gives up after
// kPatience and reports an error.
//
// What the real library rejects or hangs on is an ERROR here (DESIGN.md section 6 lists the rule behind each): a receive
// whose peer posted no send in the same round, a send nobody received by the end of the round, a size mismatch of a
// matched pair, a rank that joins a communicator twice, an unbalanced ncclGroupEnd, a point-to-point call that names
// the caller itself or a rank outside the communicator, an all-gather whose send buffer overlaps the receive buffer
// anywhere but at its own slot, an all-reduce whose buffers overlap without being equal, a reduction the stand-in does
// not model.  An error is sticky for the communicator's world: every later call of every rank fails too (the real
// library would have hung them).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct World;
struct Message {
    const void* ptr;
    size_t bytes;
    unsigned long id;     // per world, to tell the sender which of its messages was taken
};
constexpr int kPatienceMs = 4000;

struct World {
    int nranks = 0, joined = 0, left = 0;
    std::vector<char> present;   // ranks that have joined
    bool broken = false;         // a usage error was seen: every later operation fails
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    std::map<std::pair<int, int>, std::deque<Message>> mailbox; // (from, to) -> sends in issue order
    unsigned long next_id = 1;
    std::map<unsigned long, int> taken;                          // message id -> 1 delivered, 2 refused (size mismatch)
    std::condition_variable mail;
    std::vector<const void*> published;                          // collectives: every rank's send buffer
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = generation;
        if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::milliseconds(4 * kPatienceMs), [&] { return generation != g; })) {
            broken = true;   // a collective some rank never entered: the real library waits for ever
            --arrived;
        }
    }
};

struct Comm {
    World* w;
    int rank;
};

std::mutex g_m;
std::map<std::string, World*> g_worlds;

struct Op {
    bool send;
    const void* sptr;
    void* rptr;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

ncclResult_t broke(World* w, ncclResult_t rc)
{
    std::lock_guard<std::mutex> lk(w->m);
    w->broken = true;
    return rc;
}

ncclResult_t run(std::vector<Op>& ops)
{
    if (ops.empty()) return ncclSuccess;
    Comm* c = ops[0].comm;
    World* w = c->w;
    for (const Op& o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t rc = ncclSuccess;
    std::vector<unsigned long> mine;
    const auto patience = std::chrono::milliseconds(kPatienceMs);
    {
        // every send of the group is posted before any receive waits: two ranks that send to each other cannot block
        std::lock_guard<std::mutex> lk(w->m);
        if (w->broken) return ncclRemoteError;
        for (const Op& o : ops) {
            if (o.comm != c) { rc = ncclInvalidUsage; continue; }                        // (one communicator per group is all the library uses)
            if (o.peer < 0 || o.peer >= w->nranks || o.peer == c->rank) { rc = ncclInvalidArgument; continue; }
            if (!o.send) continue;
            const unsigned long id = w->next_id++;
            w->mailbox[{ c->rank, o.peer }].push_back({ o.sptr, o.bytes, id });
            mine.push_back(id);
        }
        w->mail.notify_all();
    }
    for (const Op& o : ops) {
        if (o.send || o.peer < 0 || o.peer >= w->nranks || o.peer == c->rank) continue;
        Message msg{};
        {
            std::unique_lock<std::mutex> lk(w->m);
            auto& q = w->mailbox[{ o.peer, c->rank }];
            if (!w->mail.wait_for(lk, patience, [&] { return !q.empty() || w->broken; }) || q.empty()) {
                rc = w->broken ? ncclRemoteError : ncclInvalidUsage;   // a receive nobody sent for: the real library waits for ever
                w->broken = true;
                w->mail.notify_all();
                continue;
            }
            msg = q.front();
            q.pop_front();
        }
        bool ok = msg.bytes == o.bytes;                                  // count / type mismatch of a matched pair
        if (!ok) rc = ncclInvalidArgument;
        // (a device-to-device hipMemcpy may return before the copy has run: the sender must not be told "taken" — and reuse
        // its buffer on a stream of its own — before the data has really left it)
        else if (hipMemcpy(o.rptr, msg.ptr, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->taken[msg.id] = ok ? 1 : 2;
            if (!ok) w->broken = true;
            w->mail.notify_all();
        }
    }
    {
        // a send returns once its receive has taken the data (the real library's send completes on the stream; here
        // the buffer may be reused as soon as this call returns).  A send nobody receives: the real library waits for ever
        std::unique_lock<std::mutex> lk(w->m);
        for (unsigned long id : mine) {
            if (!w->mail.wait_for(lk, patience, [&] { return w->taken.count(id) || w->broken; }) || !w->taken.count(id)) {
                if (rc == ncclSuccess) rc = w->broken ? ncclRemoteError : ncclInvalidUsage;
                w->broken = true;
                w->mail.notify_all();
                continue;
            }
            if (w->taken[id] == 2 && rc == ncclSuccess) rc = ncclInvalidArgument;
            w->taken.erase(id);
        }
        if (rc != ncclSuccess) { w->broken = true; w->mail.notify_all(); }
    }
    return rc;
}

bool overlap(const void* a, size_t na, const void* b, size_t nb)
{
    const char *x = static_cast<const char*>(a), *y = static_cast<const char*>(b);
    return x < y + nb && y < x + na;
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::random_device rd;
    std::memset(id, 0, sizeof *id);
    for (int k = 0; k < 16; ++k) id->internal[k] = static_cast<char>('a' + rd() % 26);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_m);
        World*& slot = g_worlds[std::string(id.internal, 16)];
        if (!slot) { slot = new World(); slot->nranks = nranks; slot->present.assign(nranks, 0); }
        w = slot;
        if (w->nranks != nranks) return ncclInvalidArgument;
        if (w->present[rank]) return ncclInvalidArgument;   // a rank joins a communicator once
        w->present[rank] = 1;
        ++w->joined;
    }
    Comm* c = new Comm{ w, rank };
    w->barrier(); // every rank has joined
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) { t_ops.clear(); return ncclInvalidUsage; }
    if (--t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    t_ops.push_back({ true, buf, nullptr, count * type_size(type), peer, reinterpret_cast<Comm*>(comm), stream });
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run(ops);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    t_ops.push_back({ false, nullptr, buf, count * type_size(type), peer, reinterpret_cast<Comm*>(comm), stream });
    if (t_depth) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run(ops);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    World* w = c->w;
    const size_t bytes = sendcount * type_size(type);
    if (t_depth) return broke(w, ncclInvalidUsage); // (the library never groups a collective)
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    // in place means sendbuff == recvbuff + rank * sendcount exactly; any other overlap is undefined in the real library
    const bool in_place = sendbuff == static_cast<const char*>(recvbuff) + c->rank * bytes;
    const bool bad = !in_place && overlap(sendbuff, bytes, recvbuff, bytes * w->nranks);
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->published.resize(w->nranks);
        w->published[c->rank] = sendbuff;
        if (bad) w->broken = true;
    }
    w->barrier();
    ncclResult_t rc = ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(w->m);
        if (w->broken) rc = bad ? ncclInvalidArgument : ncclRemoteError;
    }
    if (rc != ncclSuccess) { w->barrier(); return rc; }
    for (int q = 0; q < w->nranks; ++q) {
        char* dst = static_cast<char*>(recvbuff) + q * bytes;
        if (dst == w->published[q]) continue; // in place
        if (hipMemcpy(dst, w->published[q], bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    }
    if (hipDeviceSynchronize() != hipSuccess) rc = ncclUnhandledCudaError; // (device-to-device copies may still be in flight)
    w->barrier();
    return rc;
}

// sum of float / double buffers and maximum of 32-bit unsigned words, through the host (tests only: a few megabytes)
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    World* w = c->w;
    const bool sum = op == ncclSum && (type == ncclFloat32 || type == ncclFloat64), umax = op == ncclMax && type == ncclUint32;
    const size_t bytes = count * type_size(type);
    const bool bad = (!sum && !umax) || t_depth || (sendbuff != recvbuff && overlap(sendbuff, bytes, recvbuff, bytes));
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->published.resize(w->nranks);
        w->published[c->rank] = sendbuff;
        if (bad) w->broken = true;
    }
    w->barrier();
    ncclResult_t rc = ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(w->m);
        if (w->broken) rc = bad ? ncclInvalidArgument : ncclRemoteError;
    }
    if (rc != ncclSuccess) { w->barrier(); w->barrier(); return rc; }
    std::vector<char> acc(bytes), one(bytes);
    for (int q = 0; q < w->nranks; ++q) { // rank order: every rank forms the same result
        if (hipMemcpy(one.data(), w->published[q], bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        if (q == 0) acc = one;
        else if (umax) for (size_t i = 0; i < count; ++i) { unsigned &a = reinterpret_cast<unsigned*>(acc.data())[i], b = reinterpret_cast<unsigned*>(one.data())[i]; if (b > a) a = b; }
        else if (type == ncclFloat32) for (size_t i = 0; i < count; ++i) reinterpret_cast<float*>(acc.data())[i] += reinterpret_cast<float*>(one.data())[i];
        else for (size_t i = 0; i < count; ++i) reinterpret_cast<double*>(acc.data())[i] += reinterpret_cast<double*>(one.data())[i];
    }
    w->barrier(); // everybody has read the inputs: in-place outputs may be written now
    if (hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    w->barrier();
    return rc;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidUsage: return "fake_rccl: invalid usage (unmatched send / receive, unbalanced group)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument (size mismatch, bad peer, overlapping buffers, unsupported reduction)";
    case ncclRemoteError: return "fake_rccl: another rank made a usage error";
    default: return "fake_rccl: error";
    }
}

} // extern "C"
