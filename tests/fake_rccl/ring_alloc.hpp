// ring_alloc.hpp — TEST INFRASTRUCTURE: where the next group's bytes go in a sender's segment of fake_rccl_shm.cpp.  Regions are
// handed out in order and given back OLDEST FIRST (a peer consumes a pair's messages in the order they were posted); the regions in
// use run from the oldest one's offset forwards — around the end of the segment — to write_off.  Plain C++ so that
// tests/native/ring_alloc_test.cpp can drive it on the host over random sequences (no two regions in use ever overlap, none leaves
// the segment, a request that fits an empty segment is never refused for ever).
#pragma once

#include <cstddef>
#include <deque>

namespace fakerccl {

struct Region {
    size_t offset, bytes;
};

struct RingAlloc {
    size_t capacity = 0, write_off = 0;
    std::deque<Region> in_use;   // oldest first

    void reset(size_t cap) { capacity = cap; write_off = 0; in_use.clear(); }
    void release_oldest()
    {
        in_use.pop_front();
        if (in_use.empty()) write_off = 0;
    }
    // a region of `need` bytes (need > 0), or false while the regions in use leave no room for it (the caller releases and retries)
    bool try_alloc(size_t need, size_t& offset)
    {
        if (need > capacity) return false;
        if (in_use.empty()) offset = 0;
        else {
            const size_t begin = in_use.front().offset;
            if (begin <= write_off) {   // in use: [begin, write_off)
                if (write_off + need <= capacity) offset = write_off;
                else if (need < begin) offset = 0;
                else return false;
            } else if (write_off + need < begin) {   // in use: [begin, end) and [0, write_off)
                offset = write_off;
            } else {
                return false;
            }
        }
        write_off = offset + need;
        in_use.push_back({ offset, need });
        return true;
    }
};

} // namespace fakerccl
