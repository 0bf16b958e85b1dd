// fake_rccl_shm.cpp — TEST INFRASTRUCTURE: the stand-in for RCCL whose ranks are PROCESSES (each with its own HIP
// context, all on one device), bound through FPIC_RCCL_LIBRARY like fake_rccl.cpp (ranks = threads of one process).
// It exists so that one GPU can run the exact process topology of a multi-GPU launch — torch.distributed.run starting N
// copies of bench.py, the unique id handed over by rank 0, one handle and one communicator per process — where the real
// RCCL refuses two ranks on one device.  Compiled as HIP: tests/fake_rccl/Makefile.
//
// Transport: host shared memory.  Ranks of a communicator meet in a control segment named after the unique id
// (<dir>/fakerccl_<id>, dir = FAKE_RCCL_SHM_DIR or /dev/shm); every ordered pair of ranks has a ring of message
// descriptors there and a data segment of its own, created (and grown) by the sender.  A send queues a device-to-host
// copy into the data segment on the CALLER'S stream; a receive queues a host-to-device copy out of it on the caller's
// stream; nothing drains a stream, so what the caller's other streams do meanwhile is not ordered with the operation
// (as with the real library).  Matching is RCCL's: the k-th send of a pair meets the k-th receive in issue order,
// operations of a group take effect at ncclGroupEnd and all its sends are posted before any of its receives waits,
// point-to-point operations involve their two ranks only; ncclAllGather / ncclAllReduce are grouped sends and receives
// between all pairs (plus, for the reduction, a kernel that adds the contributions in rank order: every rank forms the
// same bits).
// What is host-synchronous, and stated: the host side of a receive returns once the peer's data has reached the shared
// segment (the real library's receive only enqueues), and a send's host side returns once the peer has MATCHED it.  No
// wait is ever placed on the device for another process, so no kernel or copy of this file can stall the GPU.
// FAKE_RCCL_DELAY_US / FAKE_RCCL_DELAY_WHERE: as in fake_rccl.cpp.
//
// Errors instead of hangs: a receive whose send never comes, a send nobody matches, a size mismatch of a matched pair, a
// peer outside the communicator or the caller itself, an unbalanced ncclGroupEnd, overlapping collective buffers, a
// reduction that is not modelled — each fails after FAKE_RCCL_PATIENCE_MS (default 30 s: processes start seconds apart)
// and leaves the world broken: every later call of every rank fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "ring_alloc.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kRing = 64;            // message descriptors in flight per ordered pair
constexpr size_t kAlign = 256;
constexpr uint32_t kMagic = 0x46524343u;

__global__ void delay_kernel(unsigned long long ticks)   // wall_clock64(): the constant 100 MHz counter
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

template <typename V, bool MAX>
__global__ void reduce_slots_kernel(const V* slots, int n, size_t count, V* out)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
        V acc = slots[i];
        for (int q = 1; q < n; ++q) {
            const V v = slots[q * count + i];
            if (MAX) acc = v > acc ? v : acc;
            else acc += v;
        }
        out[i] = acc;
    }
}

struct Settings {
    unsigned long long delay_ticks = 0;
    int delay_where = 3;
    int patience_ms = 30000;
    std::string dir = "/dev/shm";
    Settings()
    {
        if (const char* d = std::getenv("FAKE_RCCL_DELAY_US")) delay_ticks = 100ull * std::strtoull(d, nullptr, 10);
        if (const char* w = std::getenv("FAKE_RCCL_DELAY_WHERE")) delay_where = std::atoi(w);
        if (const char* p = std::getenv("FAKE_RCCL_PATIENCE_MS")) patience_ms = std::max(100, std::atoi(p));
        if (const char* s = std::getenv("FAKE_RCCL_SHM_DIR")) dir = s;
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

// ---- what lives in shared memory (zero-filled when the segment is created)
enum : uint32_t { E_FREE = 0, E_POSTED = 1, E_READY = 2, E_CONSUMED = 3 };
struct Entry {
    std::atomic<uint32_t> state;    // E_*: posted by the sender's call, ready when its copy has run, consumed when the receiver's copy has run
    std::atomic<uint32_t> matched;  // 0, 1 = the receive took it, 2 = the receive refused it (size mismatch)
    std::atomic<uint64_t> seq;      // 1 + index of the message among the pair's sends
    uint64_t bytes, offset;
    uint32_t gen, pad;
};
struct Pair {
    Entry ring[kRing];
};
struct Control {
    std::atomic<uint32_t> magic, nranks, joined, left, broken;
    std::atomic<uint32_t> present[kMaxRanks];
    Pair pairs[kMaxRanks][kMaxRanks]; // [from][to]
};

struct Mapping {
    void* base = nullptr;
    size_t bytes = 0;
    bool registered = false;
};

bool map_file(const std::string& path, size_t bytes, bool create, Mapping& out)
{
    int fd = open(path.c_str(), create ? O_RDWR | O_CREAT : O_RDWR, 0600);
    if (fd < 0) return false;
    if (create) {
        struct stat st{};
        if (fstat(fd, &st) != 0 || (static_cast<size_t>(st.st_size) < bytes && ftruncate(fd, static_cast<off_t>(bytes)) != 0)) { close(fd); return false; }
    } else {
        struct stat st{};
        if (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < bytes) { close(fd); return false; }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return false;
    out.base = p;
    out.bytes = bytes;
    return true;
}

void unmap(Mapping& m)
{
    if (!m.base) return;
    if (m.registered) (void)hipHostUnregister(m.base);
    munmap(m.base, m.bytes);
    m = Mapping{};
}

// pinned, so that the copies to and from the segment are truly asynchronous; if the driver refuses to pin a shared
// mapping the copies still run in stream order, only the host call may block for them
void pin(Mapping& m)
{
    m.registered = hipHostRegister(m.base, m.bytes, hipHostRegisterDefault) == hipSuccess;
    if (!m.registered) {
        (void)hipGetLastError();
        static bool told = false;
        if (!told) { told = true; std::fprintf(stderr, "fake_rccl_shm: hipHostRegister refused a shared segment; copies through it may block the host\n"); }
    }
}

struct Outstanding {                 // one group's messages to a peer: one region of the segment (ring.in_use, same order)
    std::vector<std::pair<Entry*, uint64_t>> entries; // (descriptor, sequence number it was posted with)
    bool consumed() const
    {
        for (const auto& [e, seq] : entries)
            if (e->seq.load(std::memory_order_acquire) == seq && e->state.load(std::memory_order_acquire) != E_CONSUMED) return false;
        return true;
    }
};
struct SendSide {                    // my data segment towards one peer
    Mapping seg;
    uint32_t gen = 0;
    fakerccl::RingAlloc ring;        // where the next group's bytes go (ring_alloc.hpp: host-tested on its own)
    std::deque<Outstanding> out;     // regions not yet known to be consumed, oldest first
    uint64_t sent = 0;
};
struct RecvSide {
    std::map<uint32_t, Mapping> seg; // generations of the peer's segment I have mapped
    uint64_t received = 0;
};

struct Pending {                     // the proxy sets `to` in *flag once `event` has completed
    hipEvent_t event;
    std::atomic<uint32_t>* flag;
    uint32_t from, to;
};

struct Comm {
    std::string id;
    int rank = 0, nranks = 0;
    Mapping control_map;
    Control* ctl = nullptr;
    SendSide tx[kMaxRanks];
    RecvSide rx[kMaxRanks];
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // proxy: turns "this event has completed" into a flag the peer process can see
    std::thread proxy;
    std::mutex pm;
    std::condition_variable pcv;
    std::deque<Pending> pending;
    bool stop = false;
    int device = 0;
};

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

std::string seg_path(const Comm* c, int from, int to, uint32_t gen)
{
    return settings().dir + "/fakerccl_" + c->id + "_" + std::to_string(from) + "_" + std::to_string(to) + "_" + std::to_string(gen);
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

template <typename F>
bool wait_until(const Comm* c, F cond, int patience_ms)
{
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (!cond()) {
        if (c->ctl->broken.load(std::memory_order_acquire)) return false;
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(patience_ms)) return false;
    }
    return true;
}

ncclResult_t broke(Comm* c, ncclResult_t rc)
{
    c->ctl->broken.store(1, std::memory_order_release);
    return rc;
}

void proxy_main(Comm* c)
{
    (void)hipSetDevice(c->device);
    std::vector<Pending> work;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(c->pm);
            if (work.empty()) c->pcv.wait(lk, [&] { return c->stop || !c->pending.empty(); });
            while (!c->pending.empty()) { work.push_back(c->pending.front()); c->pending.pop_front(); }
            if (c->stop && work.empty()) return;
        }
        bool fired = false;
        for (size_t i = 0; i < work.size();) {
            const hipError_t q = hipEventQuery(work[i].event);
            if (q == hipErrorNotReady) { ++i; continue; }
            if (q != hipSuccess) { (void)hipGetLastError(); c->ctl->broken.store(1, std::memory_order_release); }
            uint32_t expect = work[i].from;
            work[i].flag->compare_exchange_strong(expect, work[i].to, std::memory_order_acq_rel);
            (void)hipEventDestroy(work[i].event);
            work[i] = work.back();
            work.pop_back();
            fired = true;
        }
        if (!fired) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

bool flag_after(Comm* c, hipStream_t stream, std::atomic<uint32_t>* flag, uint32_t from, uint32_t to)
{
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventRecord(e, stream) != hipSuccess) { (void)hipEventDestroy(e); return false; }
    {
        std::lock_guard<std::mutex> lk(c->pm);
        c->pending.push_back({ e, flag, from, to });
    }
    c->pcv.notify_one();
    return true;
}

// room for the `need` bytes of ONE GROUP's messages to `peer` in my segment towards it (all of them at once: a message that
// waited for room behind another message of its own group would wait for a receive the peer posts only after ITS sends —
// two ranks that are each other's only neighbours would wait for each other).  A ring of regions reclaimed oldest first; a
// group the segment cannot hold gets a new generation of the segment once everything in the old one has been consumed
// (which needs the peer's device only: its receives for earlier groups were posted before it came here).
bool reserve(Comm* c, int peer, size_t need, size_t& offset)
{
    SendSide& s = c->tx[peer];
    const int patience = settings().patience_ms;
    auto reclaim = [&] {
        while (!s.out.empty() && s.out.front().consumed()) { s.out.pop_front(); s.ring.release_oldest(); }
    };
    reclaim();
    if (!s.seg.base || need > s.seg.bytes) {
        if (!wait_until(c, [&] { reclaim(); return s.out.empty(); }, patience)) return false;
        if (s.seg.base) { unlink(seg_path(c, c->rank, peer, s.gen).c_str()); unmap(s.seg); }
        ++s.gen;
        const size_t cap = std::max<size_t>(size_t(1) << 20, 2 * need);
        if (!map_file(seg_path(c, c->rank, peer, s.gen), cap, true, s.seg)) return false;
        pin(s.seg);
        s.ring.reset(cap);
    }
    if (!wait_until(c, [&] { reclaim(); return s.ring.try_alloc(need, offset); }, patience)) return false;
    s.out.push_back({});
    return true;
}

void* peer_segment(Comm* c, int peer, uint32_t gen, size_t upto)
{
    RecvSide& r = c->rx[peer];
    auto it = r.seg.find(gen);
    if (it != r.seg.end() && it->second.bytes >= upto) return it->second.base;
    // older generations are never used again once a newer one is seen (the sender emptied them first)
    for (auto old = r.seg.begin(); old != r.seg.end();) {
        if (old->first < gen) { unmap(old->second); old = r.seg.erase(old); }
        else ++old;
    }
    const std::string path = seg_path(c, peer, c->rank, gen);
    struct stat st{};
    if (stat(path.c_str(), &st) != 0 || static_cast<size_t>(st.st_size) < upto) return nullptr;
    Mapping m;
    if (!map_file(path, static_cast<size_t>(st.st_size), false, m)) return nullptr;
    pin(m);
    r.seg[gen] = m;
    return m.base;
}

ncclResult_t run(std::vector<Op>& ops)
{
    if (ops.empty()) return ncclSuccess;
    Comm* c = ops[0].comm;
    Control* ctl = c->ctl;
    if (ctl->broken.load(std::memory_order_acquire)) return ncclRemoteError;
    const int patience = settings().patience_ms;
    ncclResult_t rc = ncclSuccess;
    std::vector<Entry*> mine;
    auto aligned = [](size_t b) { return (b + kAlign - 1) / kAlign * kAlign; };
    size_t group_bytes[kMaxRanks] = {}, next_off[kMaxRanks] = {};
    int group_msgs[kMaxRanks] = {};
    for (const Op& o : ops) {
        if (o.comm != c) { rc = ncclInvalidUsage; continue; }
        if (o.peer < 0 || o.peer >= c->nranks || o.peer == c->rank) { rc = ncclInvalidArgument; continue; }
        if (o.send) { group_bytes[o.peer] += aligned(o.bytes); ++group_msgs[o.peer]; }
    }
    if (rc != ncclSuccess) return broke(c, rc);
    for (int q = 0; q < c->nranks; ++q) {
        if (group_msgs[q] > kRing) return broke(c, ncclInvalidUsage);   // (more messages to one peer in one group than descriptors)
        if (group_msgs[q] && !reserve(c, q, group_bytes[q], next_off[q])) return broke(c, ncclSystemError);
    }
    // every send of the group is posted before any receive waits: two ranks that send to each other cannot block
    for (const Op& o : ops) {
        if (!o.send) continue;
        SendSide& s = c->tx[o.peer];
        Entry* e = &ctl->pairs[c->rank][o.peer].ring[s.sent % kRing];
        // the descriptor's previous use (kRing messages ago) must be over
        if (!wait_until(c, [&] { const uint32_t st = e->state.load(std::memory_order_acquire); return st == E_FREE || st == E_CONSUMED; }, patience)) { rc = ncclInvalidUsage; break; }
        const size_t off = next_off[o.peer];
        next_off[o.peer] += aligned(o.bytes);
        const uint64_t seq = ++s.sent;
        e->bytes = o.bytes; e->offset = off; e->gen = s.gen;
        e->matched.store(0, std::memory_order_relaxed);
        e->state.store(E_POSTED, std::memory_order_relaxed);
        e->seq.store(seq, std::memory_order_release);
        s.out.back().entries.push_back({ e, seq });
        bool good = hipMemcpyAsync(static_cast<char*>(s.seg.base) + off, o.sptr, o.bytes, hipMemcpyDeviceToHost, o.stream) == hipSuccess;
        good = good && flag_after(c, o.stream, &e->state, E_POSTED, E_READY);
        delay_on(o.stream, 2);
        if (!good) { rc = ncclUnhandledCudaError; break; }
        mine.push_back(e);
    }
    if (rc != ncclSuccess) return broke(c, rc);
    for (const Op& o : ops) {
        if (o.send) continue;
        RecvSide& r = c->rx[o.peer];
        const uint64_t seq = r.received + 1;
        Entry* e = &ctl->pairs[o.peer][c->rank].ring[(seq - 1) % kRing];
        if (!wait_until(c, [&] { return e->seq.load(std::memory_order_acquire) == seq; }, patience)) { rc = ctl->broken.load() ? ncclRemoteError : ncclInvalidUsage; break; } // a receive nobody sent for
        r.received = seq;
        if (e->bytes != o.bytes) { e->matched.store(2, std::memory_order_release); rc = ncclInvalidArgument; break; }
        e->matched.store(1, std::memory_order_release);
        // the peer's copy into the segment has run (its stream reached the send): from here on everything is queued
        if (!wait_until(c, [&] { return e->state.load(std::memory_order_acquire) >= E_READY; }, patience)) { rc = ncclRemoteError; break; }
        const char* base = static_cast<const char*>(peer_segment(c, o.peer, e->gen, e->offset + e->bytes));
        if (!base) { rc = ncclSystemError; break; }
        delay_on(o.stream, 1);
        bool good = hipMemcpyAsync(o.rptr, base + e->offset, o.bytes, hipMemcpyHostToDevice, o.stream) == hipSuccess;
        good = good && flag_after(c, o.stream, &e->state, E_READY, E_CONSUMED);
        delay_on(o.stream, 2);
        if (!good) { rc = ncclUnhandledCudaError; break; }
    }
    if (rc != ncclSuccess) return broke(c, rc);
    // a send nobody matches: the real library waits for ever
    for (Entry* e : mine) {
        if (!wait_until(c, [&] { return e->matched.load(std::memory_order_acquire) != 0; }, patience)) { rc = ctl->broken.load() ? ncclRemoteError : ncclInvalidUsage; break; }
        if (e->matched.load() == 2) { rc = ncclInvalidArgument; break; }
    }
    if (rc != ncclSuccess) return broke(c, rc);
    return ncclSuccess;
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
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->id.assign(id.internal, 16);
    for (char ch : c->id)
        if (ch < 'a' || ch > 'z') { delete c; return ncclInvalidArgument; }   // (an id this library did not make)
    c->rank = rank; c->nranks = nranks;
    (void)hipGetDevice(&c->device);
    if (!map_file(settings().dir + "/fakerccl_" + c->id, sizeof(Control), true, c->control_map)) { delete c; return ncclSystemError; }
    c->ctl = static_cast<Control*>(c->control_map.base);
    Control* ctl = c->ctl;
    uint32_t zero = 0;
    if (ctl->nranks.compare_exchange_strong(zero, static_cast<uint32_t>(nranks)) == false && zero != static_cast<uint32_t>(nranks)) { unmap(c->control_map); delete c; return ncclInvalidArgument; }
    if (ctl->present[rank].exchange(1)) { unmap(c->control_map); delete c; return ncclInvalidArgument; } // a rank joins a communicator once
    ctl->magic.store(kMagic);
    ctl->joined.fetch_add(1);
    if (!wait_until(c, [&] { return ctl->joined.load(std::memory_order_acquire) >= static_cast<uint32_t>(nranks); }, 4 * settings().patience_ms)) {
        ctl->broken.store(1);
        unmap(c->control_map); delete c;
        return ncclSystemError;
    }
    c->proxy = std::thread(proxy_main, c);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();   // my copies have run: the proxy's queue empties
    {
        std::lock_guard<std::mutex> lk(c->pm);
        c->stop = true;
    }
    c->pcv.notify_one();
    if (c->proxy.joinable()) c->proxy.join();
    // what I sent has been read (or the world is broken and nobody will)
    for (int q = 0; q < c->nranks; ++q) {
        SendSide& s = c->tx[q];
        (void)wait_until(c, [&] {
            while (!s.out.empty() && s.out.front().consumed()) s.out.pop_front();
            return s.out.empty();
        }, settings().patience_ms / 4);
        if (s.seg.base) { unlink(seg_path(c, c->rank, q, s.gen).c_str()); unmap(s.seg); }
        for (auto& kv : c->rx[q].seg) unmap(kv.second);
    }
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->ctl->left.fetch_add(1) + 1 == static_cast<uint32_t>(c->nranks)) unlink((settings().dir + "/fakerccl_" + c->id).c_str());
    unmap(c->control_map);
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
    const size_t bytes = sendcount * type_size(type);
    if (t_depth) return broke(c, ncclInvalidUsage); // (the library never groups a collective)
    const bool in_place = sendbuff == static_cast<const char*>(recvbuff) + c->rank * bytes;
    if (!in_place && overlap(sendbuff, bytes, recvbuff, bytes * c->nranks)) return broke(c, ncclInvalidArgument);
    std::vector<Op> ops;
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank) continue;
        ops.push_back({ true, sendbuff, nullptr, bytes, q, c, stream });
        ops.push_back({ false, nullptr, static_cast<char*>(recvbuff) + q * bytes, bytes, q, c, stream });
    }
    if (c->ctl->broken.load()) return ncclRemoteError;
    if (!in_place && hipMemcpyAsync(static_cast<char*>(recvbuff) + c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return broke(c, ncclUnhandledCudaError);
    return run(ops);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    const bool sum = op == ncclSum && (type == ncclFloat32 || type == ncclFloat64), umax = op == ncclMax && type == ncclUint32;
    const size_t bytes = count * type_size(type);
    if ((!sum && !umax) || t_depth || (sendbuff != recvbuff && overlap(sendbuff, bytes, recvbuff, bytes))) return broke(c, ncclInvalidArgument);
    if (c->ctl->broken.load()) return ncclRemoteError;
    if (c->scratch_bytes < bytes * c->nranks) {
        if (c->scratch) { (void)hipDeviceSynchronize(); (void)hipFree(c->scratch); }
        c->scratch_bytes = 0;
        if (hipMalloc(&c->scratch, bytes * c->nranks) != hipSuccess) return broke(c, ncclUnhandledCudaError);
        c->scratch_bytes = bytes * c->nranks;
    }
    char* slots = static_cast<char*>(c->scratch);
    std::vector<Op> ops;
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank) continue;
        ops.push_back({ true, sendbuff, nullptr, bytes, q, c, stream });
        ops.push_back({ false, nullptr, slots + q * bytes, bytes, q, c, stream });
    }
    if (hipMemcpyAsync(slots + c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return broke(c, ncclUnhandledCudaError);
    if (ncclResult_t rc = run(ops)) return rc;
    // every contribution sits in its rank's slot; the output (possibly the send buffer itself: its copies were queued above)
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((count + 255) / 256, 1024));
    if (umax) reduce_slots_kernel<unsigned, true><<<grid, 256, 0, stream>>>(reinterpret_cast<const unsigned*>(slots), c->nranks, count, static_cast<unsigned*>(recvbuff));
    else if (type == ncclFloat32) reduce_slots_kernel<float, false><<<grid, 256, 0, stream>>>(reinterpret_cast<const float*>(slots), c->nranks, count, static_cast<float*>(recvbuff));
    else reduce_slots_kernel<double, false><<<grid, 256, 0, stream>>>(reinterpret_cast<const double*>(slots), c->nranks, count, static_cast<double*>(recvbuff));
    if (hipGetLastError() != hipSuccess) return broke(c, ncclUnhandledCudaError);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidUsage: return "fake_rccl_shm: invalid usage (unmatched send / receive, unbalanced group)";
    case ncclInvalidArgument: return "fake_rccl_shm: invalid argument (size mismatch, bad peer, overlapping buffers, unsupported reduction)";
    case ncclRemoteError: return "fake_rccl_shm: another rank failed or made a usage error";
    case ncclSystemError: return "fake_rccl_shm: a shared segment could not be created, grown or mapped";
    default: return "fake_rccl_shm: error";
    }
}

} // extern "C"
