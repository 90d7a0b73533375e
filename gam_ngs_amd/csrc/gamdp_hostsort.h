// libgamdp host side: the launch planner's sort (longest task first).  Standard library + the host pool only, so that
// tests/test_hostpool.py can build it with ThreadSanitizer / ASan on a host without HIP (tests/native/hostsort_test.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "gamdp_hostpool.h"

namespace gamdp {

// ids (ascending on entry) -> stable order of decreasing key[id]; LSD radix sort (keys < 2^40: cells of a task).
// Big batches (round 6): every pass on the host pool -- P chunks of the array histogram their digits, one thread turns the P x BINS
// counts into offsets (bins in descending digit order, chunks in order within a bin: stable), the chunks scatter.  Serially the three
// passes over 100 000 ids were 1.2 ms in front of every launch of a driver-shaped batch.
// tmp_buf / count_buf: scratch the caller keeps between calls (nullptr: allocated here)
template <class K>
inline void sort_by_key_desc(std::vector<uint32_t>& ids, const std::vector<K>& key, std::vector<uint32_t>* tmp_buf = nullptr, std::vector<size_t>* count_buf = nullptr)
{
    const size_t n = ids.size();
    if (n < 2) return;
    if (n < 4096) {
        std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
        return;
    }
    const size_t P = n >= 32768 ? 16 : 1;
    auto chunk = [&](size_t c) { return n * c / P; };
    auto on_chunks = [&](auto&& fn) {   // fn(c) for every chunk
        if (P == 1) { fn((size_t)0); return; }
        auto body = [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; c++) fn(c); };
        HostPool::get().run(P, body);
    };
    std::vector<uint64_t> cmax(P, 0);
    on_chunks([&](size_t c) { uint64_t m = 0; for (size_t k = chunk(c); k < chunk(c + 1); k++) m = std::max<uint64_t>(m, (uint64_t)key[ids[k]]); cmax[c] = m; });
    uint64_t kmax = 0;
    for (uint64_t m : cmax) kmax = std::max(kmax, m);
    // as few passes as digits of at most 10 bits allow, the bits spread evenly over them (15-bit keys: 2 passes of 8).  Wider digits cost
    // more than the passes they save: one thread walks the P x BINS counts between histogram and scatter (4 096 bins: 0.15 ms a pass)
    int nbits = 0;
    while (nbits < 64 && ((uint64_t)kmax >> nbits) != 0) nbits++;
    if (nbits == 0) return;   // every key 0: the order stays
    const int passes = (nbits + 9) / 10, BITS = (nbits + passes - 1) / passes, BINS = 1 << BITS;
    std::vector<uint32_t> tmp_own;
    std::vector<size_t> count_own;
    std::vector<uint32_t>& tmp = tmp_buf ? *tmp_buf : tmp_own;
    std::vector<size_t>& count = count_buf ? *count_buf : count_own;
    tmp.resize(n);
    if (count.size() < P * (size_t)BINS) count.resize(P * (size_t)BINS);
    std::vector<uint32_t>*src = &ids, *dst = &tmp;
    for (unsigned shift = 0; shift < 64 && ((uint64_t)kmax >> shift) != 0; shift += (unsigned)BITS) {
        const uint32_t* const sp = src->data();
        uint32_t* const dp = dst->data();
        on_chunks([&](size_t c) {
            size_t* const h = count.data() + c * BINS;
            std::fill(h, h + BINS, (size_t)0);
            for (size_t k = chunk(c); k < chunk(c + 1); k++) h[(BINS - 1) - (((uint64_t)key[sp[k]] >> shift) & (uint64_t)(BINS - 1))]++;   // inverted digit: descending order
        });
        size_t run = 0;
        for (int d = 0; d < BINS; d++)
            for (size_t c = 0; c < P; c++) { const size_t v = count[c * BINS + d]; count[c * BINS + d] = run; run += v; }
        on_chunks([&](size_t c) {
            size_t* const h = count.data() + c * BINS;
            for (size_t k = chunk(c); k < chunk(c + 1); k++) { const uint32_t i = sp[k]; dp[h[(BINS - 1) - (((uint64_t)key[i] >> shift) & (uint64_t)(BINS - 1))]++] = i; }
        });
        std::swap(src, dst);
    }
    if (src != &ids) ids.swap(tmp);
}

}  // namespace gamdp
