// Test of the launch planner's sort (gam_ngs_amd/csrc/gamdp_hostsort.h) under ThreadSanitizer / ASan: built and run by
// tests/test_hostpool.py.  sort_by_key_desc must return the ids in stable order of decreasing key -- exactly what
// std::stable_sort gives -- at every size class (the std::stable_sort path, one chunk, sixteen chunks on the host pool), for keys
// of one, two and four digits, with many ties, and from several callers at once (the contexts of a gamdp_multi call).
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "gamdp_hostsort.h"

static uint64_t rnd(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

static long check(size_t n_keys, size_t n_ids, uint64_t mask, uint64_t seed)
{
    std::vector<uint64_t> key(n_keys);
    for (auto& k : key) k = rnd(seed) & mask;
    // ids: an ascending subset of the keys' indices (the tasks of one kernel group)
    std::vector<uint32_t> ids;
    for (size_t i = 0; i < n_keys && ids.size() < n_ids; i++) if ((rnd(seed) & 3) != 0 || n_keys - i <= n_ids - ids.size()) ids.push_back((uint32_t)i);
    std::vector<uint32_t> want = ids;
    std::stable_sort(want.begin(), want.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
    gamdp::sort_by_key_desc(ids, key);
    return ids == want ? 0 : 1;
}

static long check32(size_t n, uint32_t mask, uint64_t seed)   // 32-bit keys (the rows of a task): the planner's key for one-band kernels
{
    std::vector<uint32_t> key(n), ids(n);
    for (size_t i = 0; i < n; i++) { key[i] = (uint32_t)rnd(seed) & mask; ids[i] = (uint32_t)i; }
    std::vector<uint32_t> want = ids;
    std::stable_sort(want.begin(), want.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
    std::vector<uint32_t> tmp;
    std::vector<size_t> cnt;
    gamdp::sort_by_key_desc(ids, key, &tmp, &cnt);   // (with the caller's scratch, as the planner calls it)
    return ids == want ? 0 : 1;
}

int main()
{
    std::atomic<long> bad{0};
    const size_t sizes[] = {0, 1, 2, 100, 4095, 4096, 20000, 32767, 32768, 100000, 250001};
    const uint64_t masks[] = {0x3ffull, 0x1fffffull, 0xffffffffffull, 0x7ull, 0x0ull};
    for (size_t n : sizes)
        for (uint64_t m : masks) bad += check(n + n / 3 + 5, n, m, 0x9E3779B97F4A7C15ull ^ (n * 31 + m));
    for (size_t n : {(size_t)5000, (size_t)40000, (size_t)100000})
        for (uint32_t m : {0x7fffu, 0x7ffffu, 0xffu, 0xffffffffu}) bad += check32(n, m, 1234567 + n + m);
    // several callers at once
    std::vector<std::thread> th;
    for (unsigned k = 0; k < 4; ++k) th.emplace_back([&, k] { for (int rep = 0; rep < 4; ++rep) bad += check(150000, 100000 + 1000 * k, 0x1ffffffffull, 77 + 13 * k + rep); });
    for (auto& t : th) t.join();
    std::printf("bad %ld\n", bad.load());
    return bad.load() == 0 ? 0 : 1;
}
