// fake_rccl.cpp — TEST INFRASTRUCTURE: an in-process stand-in for the ten RCCL entry points libfusionpic.so binds
// (csrc/fpic_dyn.cpp), so that the library's RCCL transport — one handle per rank, grouped ncclSend/ncclRecv of the
// decomposition, ncclAllGather / ncclAllReduce — can be driven by N threads of ONE process on ONE GPU, where the real
// RCCL refuses two ranks on one device.  Bound through FPIC_RCCL_LIBRARY by tests/test_gpu_fake_rccl.py only.
// (Ranks that are PROCESSES: fake_rccl_shm.cpp.)  Compiled as HIP (it holds two small kernels): tests/fake_rccl/Makefile.
//
// Semantics kept: ranks of a communicator meet by its unique id; between a pair of ranks the k-th send matches the k-th
// receive in issue order; operations of a group take effect at ncclGroupEnd; point-to-point operations involve their
// two ranks only (a rank with nothing to send or receive in a round takes no part in it: no barrier over the world);
// collectives involve every rank.
//
// Two ways of completing an operation (FAKE_RCCL_MODE):
//   stream (default since round 5) — STREAM-ORDERED, as the real library: a call only enqueues.  The data of a send is
//       read after everything queued before the call on the SENDER'S stream (an event recorded there), the receive is a
//       device-to-device copy queued on the RECEIVER'S stream behind that event, and what the sender queues after the
//       call on its stream waits for that copy (an event recorded on the receiver's stream) — so a buffer is never
//       reused before it has been read.  Nothing else is ordered: the caller's other streams run free, and a dependency
//       the library forgot between them (part-1 push -> ghost exchange on the communicator's stream -> interior push ->
//       join, fes_api.hip comm_fork / comm_join) shows as wrong bits.  The host side of a call still waits for the peer's
//       CALL (not for its data): the k-th receive needs to know the k-th send's address and event.
//       FAKE_RCCL_DELAY_US = n puts a spinning kernel of n microseconds (1) in front of every copy on the receiving
//       stream and / or (2) behind every operation on the stream it was issued on (FAKE_RCCL_DELAY_WHERE = bit mask,
//       default 3): arrivals come late and whatever the caller queues next on that stream starts late, which widens
//       every window a missing dependency leaves.
//   sync — rounds 2-4: the caller's stream is drained first and the copy is a blocking device-to-device copy, which is
//       stricter than stream-ordered execution (it cannot show a missing dependency).
// Where the real library would wait for ever — a receive whose send never comes, a send nobody receives — the stand-in
// gives up after kPatience and reports an error.
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
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;

// wall_clock64(): the constant 100 MHz counter
__global__ void delay_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct Sources {
    const void* p[kMaxRanks];
};

// out[i] = in_0[i] (+ or max) in_1[i] ... in rank order: every rank forms the same bits
template <typename V, bool MAX>
__global__ void reduce_kernel(Sources src, int n, size_t count, V* out)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
        V acc = static_cast<const V*>(src.p[0])[i];
        for (int q = 1; q < n; ++q) {
            const V v = static_cast<const V*>(src.p[q])[i];
            if (MAX) acc = v > acc ? v : acc;
            else acc += v;
        }
        out[i] = acc;
    }
}

struct Settings {
    bool stream = true;
    unsigned long long delay_ticks = 0;
    int delay_where = 3;
    Settings()
    {
        if (const char* m = std::getenv("FAKE_RCCL_MODE")) stream = std::strcmp(m, "sync") != 0;
        if (const char* d = std::getenv("FAKE_RCCL_DELAY_US")) delay_ticks = 100ull * std::strtoull(d, nullptr, 10);
        if (const char* w = std::getenv("FAKE_RCCL_DELAY_WHERE")) delay_where = std::atoi(w);
    }
};
const Settings& settings()
{
    static Settings s;
    return s;
}

void delay_on(hipStream_t s, int where)
{
    const Settings& cfg = settings();
    if (cfg.delay_ticks && (cfg.delay_where & where)) delay_kernel<<<1, 1, 0, s>>>(cfg.delay_ticks);
}

struct World;
struct Message {
    const void* ptr;
    size_t bytes;
    unsigned long id;     // per world, to tell the sender which of its messages was taken
    hipEvent_t ready;     // stream mode: recorded on the sender's stream when the send was issued
};
struct Taken {
    int how;              // 1 delivered, 2 refused (size mismatch)
    hipEvent_t done;      // stream mode: recorded on the receiver's stream behind the copy
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
    std::map<unsigned long, Taken> taken;
    std::condition_variable mail;
    std::vector<const void*> published;                          // collectives: every rank's send buffer
    std::vector<hipEvent_t> pub_ready, pub_done;                 // stream mode: ... is ready / has been read by this rank
    // events are taken from a ring and never destroyed while the world lives: a wait queued on a stream holds the state
    // the event had when the wait was queued, so recording it again later does not disturb the earlier wait
    std::vector<hipEvent_t> ring;
    size_t ring_next = 0;
    hipEvent_t event()
    {
        if (ring.size() < 4096) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ring.push_back(e);
            return e;
        }
        return ring[ring_next++ % ring.size()];
    }
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
    void* scratch = nullptr;   // stream mode: the all-reduce's result before every rank has read the inputs
    size_t scratch_bytes = 0;
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
    const bool ordered = settings().stream;
    Comm* c = ops[0].comm;
    World* w = c->w;
    if (!ordered)
        for (const Op& o : ops)
            if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t rc = ncclSuccess;
    std::vector<std::pair<unsigned long, hipStream_t>> mine;
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
            hipEvent_t ready = nullptr;
            if (ordered) {   // what was queued on the sender's stream before this call is what the message holds
                ready = w->event();
                if (!ready || hipEventRecord(ready, o.stream) != hipSuccess) { rc = ncclUnhandledCudaError; continue; }
            }
            w->mailbox[{ c->rank, o.peer }].push_back({ o.sptr, o.bytes, id, ready });
            mine.push_back({ id, o.stream });
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
        hipEvent_t done = nullptr;
        if (!ok) rc = ncclInvalidArgument;
        else if (ordered) {
            // behind the sender's event, on MY stream; the sender's stream will wait for `done`
            bool good = hipStreamWaitEvent(o.stream, msg.ready, 0) == hipSuccess;
            delay_on(o.stream, 1);
            good = good && hipMemcpyAsync(o.rptr, msg.ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) == hipSuccess;
            {
                std::lock_guard<std::mutex> lk(w->m);
                done = w->event();
            }
            good = good && done && hipEventRecord(done, o.stream) == hipSuccess;
            delay_on(o.stream, 2);
            if (!good) rc = ncclUnhandledCudaError;
        }
        // (a device-to-device hipMemcpy may return before the copy has run: the sender must not be told "taken" — and reuse
        // its buffer on a stream of its own — before the data has really left it)
        else if (hipMemcpy(o.rptr, msg.ptr, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->taken[msg.id] = { ok ? 1 : 2, done };
            if (!ok) w->broken = true;
            w->mail.notify_all();
        }
    }
    {
        // a send returns once its receive has taken the data (sync), or once the receive has been QUEUED and the sender's
        // stream told to wait for it (stream).  A send nobody receives: the real library waits for ever
        std::unique_lock<std::mutex> lk(w->m);
        for (const auto& [id, stream] : mine) {
            if (!w->mail.wait_for(lk, patience, [&, id = id] { return w->taken.count(id) || w->broken; }) || !w->taken.count(id)) {
                if (rc == ncclSuccess) rc = w->broken ? ncclRemoteError : ncclInvalidUsage;
                w->broken = true;
                w->mail.notify_all();
                continue;
            }
            const Taken t = w->taken[id];
            if (t.how == 2 && rc == ncclSuccess) rc = ncclInvalidArgument;
            if (t.how == 1 && ordered) {
                if (hipStreamWaitEvent(stream, t.done, 0) != hipSuccess && rc == ncclSuccess) rc = ncclUnhandledCudaError;
                delay_on(stream, 2);
            }
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

// stream mode, collectives.  Phase 1: every rank publishes its send buffer and an event recorded on its stream
// (barrier); phase 2: every rank queues — behind all those events — its own copies or its reduction, and publishes an
// event behind them (barrier); phase 3: every rank's stream waits for everybody's phase-2 event, so that no send buffer
// is rewritten while somebody still reads it.
ncclResult_t publish(Comm* c, const void* sendbuff, hipStream_t stream, bool bad)
{
    World* w = c->w;
    hipEvent_t ready;
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->published.resize(w->nranks);
        w->pub_ready.resize(w->nranks);
        w->pub_done.resize(w->nranks);
        w->published[c->rank] = sendbuff;
        ready = w->pub_ready[c->rank] = w->event();
        if (bad || !ready) w->broken = true;
    }
    if (ready && hipEventRecord(ready, stream) != hipSuccess) broke(w, ncclUnhandledCudaError);
    w->barrier();
    std::lock_guard<std::mutex> lk(w->m);
    return w->broken ? (bad ? ncclInvalidArgument : ncclRemoteError) : ncclSuccess;
}

ncclResult_t conclude(Comm* c, hipStream_t stream, ncclResult_t rc)
{
    World* w = c->w;
    hipEvent_t done;
    {
        std::lock_guard<std::mutex> lk(w->m);
        done = w->pub_done[c->rank] = w->event();
    }
    if (!done || hipEventRecord(done, stream) != hipSuccess) rc = ncclUnhandledCudaError;
    if (rc != ncclSuccess) broke(w, rc);
    w->barrier();
    for (int q = 0; q < w->nranks; ++q)
        if (q != c->rank && w->pub_done[q] && hipStreamWaitEvent(stream, w->pub_done[q], 0) != hipSuccess) rc = ncclUnhandledCudaError;
    delay_on(stream, 2);
    w->barrier(); // (the published tables may be overwritten by the next collective)
    std::lock_guard<std::mutex> lk(w->m);
    return w->broken && rc == ncclSuccess ? ncclRemoteError : rc;
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
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
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
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c && c->scratch) { (void)hipDeviceSynchronize(); (void)hipFree(c->scratch); }
    delete c;
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
    // in place means sendbuff == recvbuff + rank * sendcount exactly; any other overlap is undefined in the real library
    const bool in_place = sendbuff == static_cast<const char*>(recvbuff) + c->rank * bytes;
    const bool bad = !in_place && overlap(sendbuff, bytes, recvbuff, bytes * w->nranks);
    if (settings().stream) {
        ncclResult_t rc = publish(c, sendbuff, stream, bad);
        if (rc != ncclSuccess) { w->barrier(); w->barrier(); return rc; }
        for (int q = 0; q < w->nranks; ++q)
            if (q != c->rank && hipStreamWaitEvent(stream, w->pub_ready[q], 0) != hipSuccess) rc = ncclUnhandledCudaError;
        delay_on(stream, 1);
        for (int q = 0; q < w->nranks; ++q) {
            char* dst = static_cast<char*>(recvbuff) + q * bytes;
            if (dst == w->published[q]) continue; // in place
            if (hipMemcpyAsync(dst, w->published[q], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = ncclUnhandledCudaError;
        }
        return conclude(c, stream, rc);
    }
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

// sum of float / double buffers and maximum of 32-bit unsigned words (stream mode: a kernel that reads every rank's
// buffer in rank order; sync mode: through the host — tests only, a few megabytes)
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    World* w = c->w;
    const bool sum = op == ncclSum && (type == ncclFloat32 || type == ncclFloat64), umax = op == ncclMax && type == ncclUint32;
    const size_t bytes = count * type_size(type);
    const bool bad = (!sum && !umax) || t_depth || (sendbuff != recvbuff && overlap(sendbuff, bytes, recvbuff, bytes));
    if (settings().stream) {
        ncclResult_t rc = publish(c, sendbuff, stream, bad);
        if (rc != ncclSuccess) { w->barrier(); w->barrier(); return rc; }
        if (c->scratch_bytes < bytes) {
            if (c->scratch) { (void)hipDeviceSynchronize(); (void)hipFree(c->scratch); }
            c->scratch_bytes = 0;
            if (hipMalloc(&c->scratch, bytes) != hipSuccess) rc = ncclUnhandledCudaError;
            else c->scratch_bytes = bytes;
        }
        for (int q = 0; q < w->nranks; ++q)
            if (q != c->rank && hipStreamWaitEvent(stream, w->pub_ready[q], 0) != hipSuccess) rc = ncclUnhandledCudaError;
        delay_on(stream, 1);
        if (rc == ncclSuccess) {
            Sources src{};
            for (int q = 0; q < w->nranks; ++q) src.p[q] = w->published[q];
            const unsigned grid = static_cast<unsigned>(std::min<size_t>((count + 255) / 256, 1024));
            if (umax) reduce_kernel<unsigned, true><<<grid, 256, 0, stream>>>(src, w->nranks, count, static_cast<unsigned*>(c->scratch));
            else if (type == ncclFloat32) reduce_kernel<float, false><<<grid, 256, 0, stream>>>(src, w->nranks, count, static_cast<float*>(c->scratch));
            else reduce_kernel<double, false><<<grid, 256, 0, stream>>>(src, w->nranks, count, static_cast<double*>(c->scratch));
            if (hipGetLastError() != hipSuccess) rc = ncclUnhandledCudaError;
        }
        // conclude(): every rank's stream waits until everybody has READ the inputs; only then may an in-place output land
        rc = conclude(c, stream, rc);
        if (rc == ncclSuccess && hipMemcpyAsync(recvbuff, c->scratch, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = ncclUnhandledCudaError;
        return rc;
    }
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
