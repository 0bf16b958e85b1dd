// ring_alloc_test.cpp — tests/fake_rccl/ring_alloc.hpp (the allocator of the shared-memory stand-in's data segments) over random
// sequences on the host: regions in use never overlap and never leave the segment, they are released oldest first, and a request
// that fits the segment succeeds once enough of the oldest regions have been released.
#include "../fake_rccl/ring_alloc.hpp"

#include <cstdio>
#include <random>
#include <vector>

int main()
{
    std::mt19937_64 rng(12345);
    long cases = 0, allocs = 0, wraps = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        fakerccl::RingAlloc a;
        const size_t cap = 256 * (4 + rng() % 200);
        a.reset(cap);
        for (int step = 0; step < 400; ++step) {
            ++cases;
            if (rng() % 3 && !a.in_use.empty()) { a.release_oldest(); continue; }
            const size_t need = 256 * (1 + rng() % (cap / 256 + 2));   // sometimes larger than the segment
            size_t off = ~size_t(0);
            int spins = 0;
            const size_t before = a.write_off;
            while (!a.try_alloc(need, off)) {
                if (need > cap) break;                      // refused for good: the caller grows the segment
                if (a.in_use.empty()) { std::printf("refused on an empty segment: need %zu cap %zu\n", need, cap); return 1; }
                a.release_oldest();
                if (++spins > 1000) { std::printf("no progress\n"); return 1; }
            }
            if (need > cap) continue;
            ++allocs;
            if (off < before && !a.in_use.empty() && a.in_use.size() > 1) ++wraps;
            if (off + need > cap) { std::printf("region [%zu, %zu) leaves the segment of %zu\n", off, off + need, cap); return 1; }
            for (size_t i = 0; i < a.in_use.size(); ++i)
                for (size_t j = i + 1; j < a.in_use.size(); ++j) {
                    const auto &x = a.in_use[i], &y = a.in_use[j];
                    if (x.offset < y.offset + y.bytes && y.offset < x.offset + x.bytes) {
                        std::printf("regions [%zu, %zu) and [%zu, %zu) overlap\n", x.offset, x.offset + x.bytes, y.offset, y.offset + y.bytes);
                        return 1;
                    }
                }
        }
    }
    std::printf("cases=%ld allocations=%ld wraps=%ld\nok\n", cases, allocs, wraps);
    return 0;
}
