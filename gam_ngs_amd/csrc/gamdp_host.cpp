// Host side of libgamdp (C ABI in include/gamdp.h): contexts, packed sequence sets, the L0 batch
// launcher and the small host functions on the path (encode / revcomp / findHits / synthetic pairs).
// The merge-block chain driver (L1) lives in gamdp_l1.cpp.
//
// There is deliberately no CPU implementation of the DP here: every alignment goes through the
// gfx950 kernels in gamdp_kernel.hip, and context creation fails without a GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gamdp.h"
#include "gamdp_dev.h"
#include "gamdp_internal.h"
#include "gamdp_hostpool.h"
#include "gamdp_hostsort.h"

namespace gamdp {

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
            return GAMDP_EHIP;                                                                    \
        }                                                                                         \
    } while (0)

// ---- diagnostics switches ------------------------------------------------------------------------
// The product build (libgamdp.so) ignores every GAMDP_DIAG_* environment variable except GAMDP_DIAG_TIMING (stderr
// timing lines, results untouched).  The switches that change which kernel path runs, or that make results unusable,
// exist only in the diagnostics build (`make diag` -> libgamdp_diag.so, compiled with -DGAMDP_DIAG), which tests and
// profiling select through GAMDP_LIB.
const Diag& diag()
{
    static const Diag d = [] {
        Diag x;
        x.timing = std::getenv("GAMDP_DIAG_TIMING") != nullptr;
#ifdef GAMDP_DIAG
        x.build = true;
        x.skip_traceback = std::getenv("GAMDP_DIAG_SKIP_TRACEBACK") != nullptr;
        x.no_dirfree = std::getenv("GAMDP_DIAG_NO_DIRFREE") != nullptr;
        x.count_mat = std::getenv("GAMDP_DIAG_COUNT_MAT") != nullptr;
        x.force_n = std::getenv("GAMDP_DIAG_FORCE_N") != nullptr;
#endif
        return x;
    }();
    return d;
}

// ---- sequence packing ---------------------------------------------------------------------------

static inline uint8_t encode_char(char ch)
{  // Nucleotide(char), nucleotide.code.hpp:47-75
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'T': case 't': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    default: return 4;
    }
}

// words needed for one padded sequence in each plane (rounded so every sequence starts 64-base aligned)
static inline u64 padded_bases(u64 len) { return ((len + 2 * (u64)SEQ_PAD_BASES + 63) / 64) * 64 + 64; }

// pack codes[0..len) into plane2/planeN starting at base offset `at` (a multiple of 32)
static void pack_into(const uint8_t* codes, u64 len, bool rc, u32* plane2, u32* planeN, u64 at)
{
    static const uint8_t comp[5] = {1, 0, 3, 2, 4};
    for (u64 i = 0; i < len; i++) {
        uint8_t c = rc ? comp[codes[len - 1 - i] > 4 ? 4 : codes[len - 1 - i]] : codes[i];
        if (c > 4) c = 4;
        const u64 k = at + i;
        if (c == 4) planeN[k >> 5] |= 1u << (k & 31);
        else plane2[k >> 4] |= (u32)c << ((k & 15) * 2);
    }
}

int SeqSet::upload(Ctx* ctx_, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, bool ascii)
{
    ctx = ctx_;
    codes.resize(n);
    this->lens.assign(lens, lens + n);
    fwd.resize(n);
    rc.assign(n, DevSeq{nullptr, nullptr});
    has_n.assign(n, 0);
    npre.assign(n, std::vector<u32>());
    u64 total = 0;
    std::vector<u64> at(n);
    for (u32 i = 0; i < n; i++) {
        if (lens[i] >= (1ull << 31) - (1ull << 20)) return GAMDP_EINVAL;  // keeps begin_a + X + 64*17 inside int32 in the kernels
        at[i] = total + SEQ_PAD_BASES;  // total stays a multiple of 64
        total += padded_bases(lens[i]);
    }
    std::vector<u32> h2(total / 16 + 4, 0), hn(total / 32 + 4, 0);
    for (u32 i = 0; i < n; i++) {
        codes[i].resize(lens[i]);
        bool anyn = false;
        for (u64 k = 0; k < lens[i]; k++) {
            uint8_t c = ascii ? encode_char((char)seqs[i][k]) : (seqs[i][k] > 4 ? (uint8_t)4 : seqs[i][k]);
            codes[i][k] = c;
            anyn |= (c == 4);
        }
        has_n[i] = anyn;
        if (anyn) {
            std::vector<u32>& pre = npre[i];
            pre.assign((size_t)(lens[i] + 255) / 256 + 1, 0);
            for (u64 k = 0; k < lens[i]; k++) pre[(size_t)(k / 256) + 1] += (codes[i][k] == 4);
            for (size_t k = 1; k < pre.size(); k++) pre[k] += pre[k - 1];
        }
        pack_into(codes[i].data(), lens[i], false, h2.data(), hn.data(), at[i]);
    }
    HIPCHK(ctx, hipMalloc(&d2, h2.size() * sizeof(u32)));
    HIPCHK(ctx, hipMalloc(&dn, hn.size() * sizeof(u32)));
    HIPCHK(ctx, hipMemcpyAsync(d2, h2.data(), h2.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dn, hn.data(), hn.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (u32 i = 0; i < n; i++) fwd[i] = DevSeq{d2 + at[i] / 16, dn + at[i] / 32};
    // the N counts per 256 bases, for the chain kernels' choice of cell call by call
    dev_npre.assign(n, nullptr);
    {
        std::vector<u32> all;
        std::vector<size_t> first(n, 0);
        for (u32 i = 0; i < n; i++) { first[i] = all.size(); all.insert(all.end(), npre[i].begin(), npre[i].end()); }
        if (!all.empty()) {
            HIPCHK(ctx, hipMalloc(&d_npre, all.size() * sizeof(u32)));
            HIPCHK(ctx, hipMemcpy(d_npre, all.data(), all.size() * sizeof(u32), hipMemcpyHostToDevice));
            for (u32 i = 0; i < n; i++) if (!npre[i].empty()) dev_npre[i] = d_npre + first[i];
        }
    }
    return 0;
}


// splitmix64 stream of synthetic pair k (same generator as gamdp_synth_pair below)
namespace {
struct SynthGen {
    uint64_t s;
    explicit SynthGen(uint64_t k) : s(0x47414DULL + k * 0xD1B54A32D192ED03ULL) { (void)next(); }
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};
uint64_t synth_pair_codes(uint64_t k, uint64_t len, uint8_t* master, uint8_t* slave)
{
    SynthGen g(k);
    for (uint64_t i = 0; i < len; i++) master[i] = (uint8_t)(g.next() >> 62);
    uint64_t n = 0;
    for (uint64_t i = 0; i < len; i++) {
        const uint64_t r = g.next(), u = r >> 40;
        if (u >= 167772) {                       // 1 % deletion below this
            uint8_t base = master[i];
            if (u < 671088) base = (uint8_t)((base + 1 + (r & 0xFFFF) % 3) & 3);  // 3 % substitution
            slave[n++] = base;
        }
        const uint64_t r2 = g.next();
        if ((r2 >> 40) < 167772) slave[n++] = (uint8_t)(r2 & 3);                 // 1 % insertion
    }
    return n;
}
}  // namespace

// Benchmark helper: generate synthetic pairs first_pair + k*stride_pairs, k < n_pairs, on host threads straight into the
// packed planes (sequence 2k = master, 2k+1 = slave) and upload them; no 1 B/base host copy is kept.
int SeqSet::upload_synth(Ctx* ctx_, uint64_t first_pair, uint64_t stride_pairs, uint32_t n_pairs, uint64_t len)
{
    ctx = ctx_;
    const u32 n = 2 * n_pairs;
    lens.assign(n, 0);
    fwd.resize(n);
    rc.assign(n, DevSeq{nullptr, nullptr});
    has_n.assign(n, 0);
    dev_npre.assign(n, nullptr);
    const u64 stride = padded_bases(len + len / 8 + 64);  // fixed slot per sequence (slave length varies)
    const u64 total = stride * n;
    std::vector<u32> h2(total / 16 + 4, 0), hn(total / 32 + 4, 0);
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    Threads pool;
    std::atomic<u32> cursor(0);
    std::atomic<int> failed(0);
    for (unsigned t = 0; t < nt; t++) {
        pool.start([&]() {
            failed |= guarded_thread_body([&]() -> int {
            std::vector<uint8_t> m(len), s(len + len / 8 + 64);
            for (;;) {
                const u32 k = cursor.fetch_add(1);
                if (k >= n_pairs) break;
                const u64 sl = synth_pair_codes(first_pair + (u64)k * stride_pairs, len, m.data(), s.data());
                lens[2 * k] = len;
                lens[2 * k + 1] = sl;
                // slots are 64-base aligned and disjoint, so threads never touch the same word
                pack_into(m.data(), len, false, h2.data(), hn.data(), stride * (2 * k) + SEQ_PAD_BASES);
                pack_into(s.data(), sl, false, h2.data(), hn.data(), stride * (2 * k + 1) + SEQ_PAD_BASES);
            }
            return 0;
            });
        });
    }
    pool.join();
    if (failed) return GAMDP_ENOMEM;
    HIPCHK(ctx, hipMalloc(&d2, h2.size() * sizeof(u32)));
    HIPCHK(ctx, hipMalloc(&dn, hn.size() * sizeof(u32)));
    HIPCHK(ctx, hipMemcpyAsync(d2, h2.data(), h2.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dn, hn.data(), hn.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (u32 i = 0; i < n; i++) fwd[i] = DevSeq{d2 + (stride * i + SEQ_PAD_BASES) / 16, dn + (stride * i + SEQ_PAD_BASES) / 32};
    return 0;
}

// make sure reverse-complement copies exist on the device for the listed sequence ids
int SeqSet::ensure_rc(const std::vector<u32>& ids, Ctx* use) const
{
    std::lock_guard<std::mutex> lock(rc_mu);
    Ctx* const ctx = use ? use : this->ctx;  // uploads go through the calling context's stream
    std::vector<u32> todo;
    for (u32 id : ids)
        if (!rc[id].p2 && std::find(todo.begin(), todo.end(), id) == todo.end()) todo.push_back(id);
    if (todo.empty()) return 0;
    u64 total = 0;
    std::vector<u64> at(todo.size());
    for (size_t k = 0; k < todo.size(); k++) {
        at[k] = total + SEQ_PAD_BASES;
        total += padded_bases(codes[todo[k]].size());
    }
    std::vector<u32> h2(total / 16 + 4, 0), hn(total / 32 + 4, 0);
    for (size_t k = 0; k < todo.size(); k++)
        pack_into(codes[todo[k]].data(), codes[todo[k]].size(), true, h2.data(), hn.data(), at[k]);
    u32 *p2 = nullptr, *pn = nullptr;
    HIPCHK(ctx, hipMalloc(&p2, h2.size() * sizeof(u32)));
    rc_allocs.push_back(p2);  // owned from here on, whatever fails next
    HIPCHK(ctx, hipMalloc(&pn, hn.size() * sizeof(u32)));
    rc_allocs.push_back(pn);
    HIPCHK(ctx, hipMemcpyAsync(p2, h2.data(), h2.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pn, hn.data(), hn.size() * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < todo.size(); k++) rc[todo[k]] = DevSeq{p2 + at[k] / 16, pn + at[k] / 32};
    return 0;
}

SeqSet::~SeqSet()
{
    if (d2) (void)hipFree(d2);
    if (dn) (void)hipFree(dn);
    if (d_npre) (void)hipFree(d_npre);
    for (u32* p : rc_allocs) (void)hipFree(p);
}

// ---- context ------------------------------------------------------------------------------------

int Ctx::init(int dev)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || dev < 0 || dev >= ndev) {
        set_error("no usable HIP device");
        return GAMDP_ENODEV;
    }
    device = dev;
    if (hipSetDevice(dev) != hipSuccess) { set_error("hipSetDevice failed"); return GAMDP_ENODEV; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return GAMDP_ENODEV; }
    n_cu = prop.multiProcessorCount;
    max_lds_per_wg = (u32)std::min<size_t>(prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536, 160u * 1024u);
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        set_error(std::string("device is ") + prop.gcnArchName + ", libgamdp is built for gfx950 only");
        return GAMDP_ENODEV;
    }
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return GAMDP_ENODEV; }
    cap_cursor = 64;
    if (hipMalloc(&d_cursor, (size_t)cap_cursor * (1 + LS_COUNT) * sizeof(u32)) != hipSuccess) { set_error("hipMalloc failed"); return GAMDP_ENOMEM; }
    return 0;
}

Ctx::~Ctx()
{
    for (Ctx* h : helpers) delete h;
    if (device >= 0) (void)hipSetDevice(device);
    flush_frees();
    if (ref_event) (void)hipEventDestroy(ref_event);
    for (auto& ev : events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (d_scratch) (void)hipFree(d_scratch);
    if (d_tasks) (void)hipFree(d_tasks);
    if (d_results) (void)hipFree(d_results);
    if (d_ops) (void)hipFree(d_ops);
    if (d_cursor) (void)hipFree(d_cursor);
    if (d_chain) (void)hipFree(d_chain);
    if (h_chain) (void)hipHostFree(h_chain);
    if (h_mirror) (void)hipHostFree(h_mirror);
    if (d_chain_scratch) (void)hipFree(d_chain_scratch);
    if (chain_stream) (void)hipStreamDestroy(chain_stream);
    if (h_tasks) (void)hipHostFree(h_tasks);
    if (h_results) (void)hipHostFree(h_results);
    if (aux_done) (void)hipEventDestroy(aux_done);
    if (aux_go) (void)hipEventDestroy(aux_go);
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

u64 Ctx::arena_budget(const bool refresh)
{
    if (arena_limit == 0 || (refresh && arena_auto)) {
        size_t fr = 0, tot = 0;
        if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
        u64 held = cap_scratch + cap_chain_scratch;
        for (const Ctx* h : helpers) held += h->cap_scratch + h->cap_chain_scratch;
        arena_limit = (u64)((double)(fr + held * sizeof(u32)) * 0.75);
        arena_auto = true;
    }
    return arena_limit;
}

void Ctx::trim_scratch()
{
    if (d_scratch && cap_scratch * sizeof(u32) > arena_call()) {
        (void)hipSetDevice(device);
        (void)hipFree(d_scratch);
        d_scratch = nullptr; cap_scratch = 0;
    }
}

template <class T>
static int grow(Ctx* ctx, T*& ptr, u64& cap, u64 need)
{
    if (need <= cap) return 0;
    if (ptr) { ctx->free_dev(ptr); ptr = nullptr; cap = 0; }
    u64 want = need + (ctx->defer_frees ? need + 1024 : need / 4);   // (a round loop beside a chain launch: see the scratch arena in Ctx::align)
    if (hipMalloc(&ptr, want * sizeof(T)) != hipSuccess) {
        if (hipMalloc(&ptr, need * sizeof(T)) != hipSuccess) {
            ctx->set_error("hipMalloc of " + std::to_string(need * sizeof(T)) + " bytes failed");
            return GAMDP_ENOMEM;
        }
        want = need;
    }
    cap = want;
    return 0;
}

// ---- task validation (same order as banded_smith_waterman.cc:90-132) -------------------------------

// words of a task's direction image in a slot of kernel `kid` (K_WIDE: the band matrix itself).  The multi-task / packed kernels keep
// three blocks more: the packed range of a wavefront ends behind its longest task, rounded up to a group, and a strip of the last group
// writes the direction words of the whole group (run_octo, run_pair)
static u64 dir_words_for(int kid, u64 X, u64 band)
{
    const u64 Y = 2 * band + 1;
    if (kid == K_WIDE) return X * Y;
    const u64 C = (u64)kernel_cols(kid), LE = (Y - 1) / C;
    const u64 extra = (kid == K_P17_CE4 || kid == K_O19_CE15 || kid == K_Q19_CE15 || kid == K_Q19_CE15_N) ? 3 : 0;
    return ((X - 1 + LE) / 16 + 1 + extra) * (u64)kernel_dir_block_words(kid);
}
// the tuned kernels take one band each (pick_kernel): 0 = any band (the generic and the wide kernels)
static u32 kernel_band(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C17_CE4_N: case K_P17_CE4: return 512;
    case K_C5_CE0: case K_C5_CE0_N: case K_O19_CE15: case K_Q19_CE15: case K_Q19_CE15_N: return 150;
    default: return 0;
    }
}

static int pick_kernel(int band, bool has_n)
{
    if (band == 512) return has_n ? K_C17_CE4_N : K_C17_CE4;
    if (band == 150) return has_n ? K_C5_CE0_N : K_C5_CE0;
    const int Y = 2 * band + 1;
    if (Y <= 2 * 64) return K_GEN_C2;
    if (Y <= 3 * 64) return K_GEN_C3;
    if (Y <= 5 * 64) return K_GEN_C5;
    if (Y <= 9 * 64) return K_GEN_C9;
    if (Y <= 17 * 64) return K_GEN_C17;
    return K_WIDE;   // band > 543: gamdp_wide.hip
}

// fn(lo, hi) over [0, n) on the process's host thread pool (gamdp_hostpool.h: HostPool); batches of a few thousand tasks are not worth a thread
template <class F>
static void parallel_for(size_t n, F fn)
{
    if (n < 8192) { fn((size_t)0, n); return; }   // (the bodies passed here only index pre-sized arrays: they do not throw)
    HostPool::get().run(n, fn);
}

// Validation in the order of banded_smith_waterman.cc:90-132 on plain numbers.  Returns GAMDP_ST_OK when the task has
// to run on the GPU (then *X_out = rows of the band matrix), otherwise the final status the reference's behaviour maps
// to; *cells_out = x_size * y_size whenever the reference got as far as sizing its matrix.
int preflight(u64 alen, u64 blen, u64 band, u64 begin_a, u64 end_a, u64 begin_b, u64 end_b, bool fs, bool fe,
              u64* X_out, u64* cells_out)
{
    return preflight_hd(alen, blen, band, begin_a, end_a, begin_b, end_b, fs, fe, X_out, cells_out);   // gamdp_dev.h: shared with the chain kernel
}

// returns GAMDP_ST_OK when the task has to run on the GPU, otherwise its final status
static int prepare_task(const ITask& it, Prepared& pr)
{
    const u64 alen_full = it.sa->lens[it.a_id], blen_full = it.sb->lens[it.b_id];
    pr.cells = 0;
    if (it.a_off > alen_full || it.b_off > blen_full) return GAMDP_ST_INVALID;
    const u64 alen = alen_full - it.a_off, blen = blen_full - it.b_off;
    const u64 band = it.band;
    const bool fs = it.force_start, fe = it.force_end;
    u64 X = 0;
    const int st = preflight(alen, blen, band, it.begin_a, it.end_a, it.begin_b, it.end_b, fs, fe, &X, &pr.cells);
    if (st != GAMDP_ST_OK) return st;
    // GAMDP_DIAG_FORCE_N (diagnostics build only): run N-free inputs through the N-aware kernels as well
    // N by window, not by contig: the DP touches a[begin_a - band .. begin_a + X - 1 + band] and b[begin_b .. begin_b + X - 1] (pos =
    // begin_a - band + x + y, banded_smith_waterman.cc:135-171), the walk stays inside them; 64 bases of margin on either side.
    // GAMDP_N_BY_CONTIG=1: the contig's flag decides, as before round 4 (A/B, and a second way through the tests).
    static const bool n_by_contig = std::getenv("GAMDP_N_BY_CONTIG") != nullptr;
    bool has_n = diag().force_n;
    if (!has_n && n_by_contig) has_n = it.sa->has_n[it.a_id] || it.sb->has_n[it.b_id];
#ifdef GAMDP_DIAG
    // fault injection (diagnostics build): windows too small by this many bases on either side -- the test of the windows must notice
    static const int64_t margin = 64 - [] { const char* e = std::getenv("GAMDP_DIAG_N_WINDOW_SHRINK"); return e ? (int64_t)std::atol(e) : (int64_t)0; }();
#else
    constexpr int64_t margin = 64;
#endif
    if (!has_n && !n_by_contig)
        has_n = it.sa->window_has_n(it.a_id, it.a_rc, it.a_off, (int64_t)it.begin_a - (int64_t)band - margin, (int64_t)it.begin_a + (int64_t)X - 1 + (int64_t)band + margin) ||
                it.sb->window_has_n(it.b_id, it.b_rc, it.b_off, (int64_t)it.begin_b - margin, (int64_t)it.begin_b + (int64_t)X - 1 + margin);
    pr.kid = pick_kernel((int)band, has_n);
    pr.dir_words = dir_words_for(pr.kid, X, band);   // (of the kernel picked here: the planner sizes a group that moves to a multi-task kernel by dir_words_for itself)
    DevTask& d = pr.dt;
    const DevSeq& da = it.a_rc ? it.sa->rc[it.a_id] : it.sa->fwd[it.a_id];
    const DevSeq& db = it.b_rc ? it.sb->rc[it.b_id] : it.sb->fwd[it.b_id];
    d.a2 = da.p2; d.an = da.pn; d.b2 = db.p2; d.bn = db.pn;
    d.a_base = (int64_t)it.a_off; d.b_base = (int64_t)it.b_off;
    d.end_a = (int64_t)std::min<u64>(it.end_a, 1ull << 40);
    d.alen = (int32_t)alen; d.blen = (int32_t)blen;
    d.begin_a = (int32_t)it.begin_a; d.begin_b = (int32_t)it.begin_b;
    d.X = (int32_t)X; d.band = (int32_t)band;
    d.flags = (fs ? TF_FORCE_START : 0u) | (fe ? TF_FORCE_END : 0u);
    d.ops_off = 0; d.ops_cap = 0;
    return GAMDP_ST_OK;
}

void fill_result(const DevResult& r, u64 cells, gamdp_result& o)
{
    std::memset(&o, 0, sizeof(o));
    o.status = (uint8_t)(r.flags >> 8);
    o.cells = cells;
    if (o.status != GAMDP_ST_OK) return;
    o.begin_a = (u64)(int64_t)r.begin_a; o.begin_b = (u64)(int64_t)r.begin_b;
    o.score = r.score;
    o.n_match = r.n_match; o.length = r.length;
    o.first_a = (u64)(int64_t)r.first_a; o.first_b = (u64)(int64_t)r.first_b;
    o.last_a = (u64)(int64_t)r.last_a; o.last_b = (u64)(int64_t)r.last_b;
    o.first_found = r.flags & 1u; o.last_found = (r.flags >> 1) & 1u;
    o.homology = o.length ? (double)(o.n_match * 100) / (double)o.length : 0.0;  // :319
}

// ---- L0 batch -----------------------------------------------------------------------------------

int Ctx::align(const TaskSrc& tasks, size_t n, gamdp_result* out, const gamdp_ops* ops)
{
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return GAMDP_EHIP; }
    if (n == 0) return 0;
    // reverse complements needed by this batch
    {
        std::vector<std::pair<const SeqSet*, std::vector<u32>>> need;
        auto add = [&](const SeqSet* s, u32 id) {
            for (auto& p : need) if (p.first == s) { p.second.push_back(id); return; }
            need.push_back({s, {id}});
        };
        // (checked in parallel: a serial pass over 100 000 tasks is 0.2 ms in front of every launch; the first offender by index is reported)
        std::atomic<size_t> bad_at{n};
        std::atomic<bool> any_rc{false};
        parallel_for(n, [&](size_t lo, size_t hi) {
            bool rc_seen = false;
            for (size_t ti = lo; ti < hi; ti++) {
                const ITask t = tasks[ti];
                const bool bad = t.band > GAMDP_MAX_BAND || t.a_id >= t.sa->lens.size() || t.b_id >= t.sb->lens.size() ||
                                 (t.a_rc && !t.sa->has_codes()) || (t.b_rc && !t.sb->has_codes());
                if (bad) { size_t cur = bad_at.load(); while (ti < cur && !bad_at.compare_exchange_weak(cur, ti)) {} break; }
                rc_seen |= t.a_rc || t.b_rc;
            }
            if (rc_seen) any_rc.store(true);
        });
        if (bad_at.load() < n) {
            const ITask t = tasks[bad_at.load()];
            if (t.band > GAMDP_MAX_BAND) { set_error("band " + std::to_string(t.band) + " exceeds GAMDP_MAX_BAND"); return GAMDP_ENOTSUP; }
            if (t.a_id >= t.sa->lens.size() || t.b_id >= t.sb->lens.size()) { set_error("sequence id out of range"); return GAMDP_EINVAL; }
            set_error("reverse complement requested on a packed-only (synthetic) sequence set"); return GAMDP_EINVAL;
        }
        if (any_rc.load()) {
            for (size_t ti = 0; ti < n; ti++) {
                const ITask t = tasks[ti];
                if (t.a_rc) add(t.sa, t.a_id);
                if (t.b_rc) add(t.sb, t.b_id);
            }
        }
        for (auto& p : need) { int rc_ = p.first->ensure_rc(p.second, this); if (rc_) return rc_; }
    }

    const bool diag_skip_tb = diag().skip_traceback, diag_timing = diag().timing, diag_no_dirfree = diag().no_dirfree,
               diag_count_mat = diag().count_mat;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // GAMDP_DIAG_TIMING: where the host side of the call spends its time, phase by phase (ms since the call began)
    std::vector<std::pair<const char*, double>> marks;
    auto mark = [&](const char* what) { if (diag_timing) marks.push_back({what, since(t_begin)}); };
    // per-batch work arrays live in the context: a fresh 50 MB vector per call costs more in page faults than the
    // preparation itself
    if (w_prep.size() < n) { w_prep.resize(n); w_status.resize(n); w_key.resize(n); w_kid.resize(n); w_rows.resize(n); }
    std::vector<Prepared>& prep = w_prep;
    std::vector<int>& prep_status = w_status;
    // what the serial steps below need of a task, side by side (round 6): its kernel (-1: settled by the pre-checks) and its rows --
    // 5 bytes per task instead of a pass over the 120-byte descriptors (12 MB per 100 000 tasks, three times over: 1.1 ms of a
    // driver-shaped batch's 4.3 ms in front of the launch)
    std::vector<int8_t>& kidv = w_kid;
    std::vector<u32>& rowsv = w_rows;
    const u32 diag_flags = (diag_skip_tb ? (u32)TF_DIAG_SKIP_TRACEBACK : 0u) | (diag_no_dirfree ? (u32)TF_NO_DIRFREE : 0u) | (diag_count_mat ? (u32)TF_DIAG_COUNT_MAT : 0u);
    const bool any_ops = ops && ops->ops_buf;
    // validation + descriptor of every task: independent per task, spread over host threads for big batches
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const int st = prep_status[i] = prepare_task(tasks[i], prep[i]);
            w_key[i] = prep[i].cells;   // the planner's sort key (and the `cells` of the result)
            if (st != GAMDP_ST_OK) {
                kidv[i] = -1; rowsv[i] = 0;
                std::memset(&out[i], 0, sizeof(out[i]));
                out[i].status = (uint8_t)st;
                out[i].cells = prep[i].cells;  // 0 unless the reference got as far as sizing its matrix
                continue;
            }
            prep[i].dt.res_idx = (u32)i;
            prep[i].dt.flags |= diag_flags;
            kidv[i] = (int8_t)prep[i].kid; rowsv[i] = (u32)prep[i].dt.X;
        }
    });
    mark("pre-checks");
    // (the per-kernel task lists and the sort's scratch keep their memory from call to call: freeing and re-allocating half a megabyte
    // per call goes through mmap / munmap, and an munmap interrupts every thread of the host pool)
    u64 rows_of[K_COUNT] = {0};   // rows of the tasks of every kernel's group
    if (w_groups.size() != (size_t)K_COUNT) w_groups.resize(K_COUNT);
    std::vector<std::vector<u32>>& groups = w_groups;
    for (auto& gv : groups) gv.clear();
    u64 ops_total = 0;
    if (any_ops) {   // edit strings: where each task's go (in task order)
        for (size_t i = 0; i < n; i++) {
            if (kidv[i] < 0 || ops->ops_cap[i] == 0) continue;
            prep[i].dt.flags |= TF_WANT_OPS;
            prep[i].dt.ops_off = ops_total;
            prep[i].dt.ops_cap = ops->ops_cap[i];
            ops_total += ops->ops_cap[i];
        }
    }
    {
        // (how many tasks and how many rows each kernel got: counted by the parts of a parallel loop -- every serial pass over 100 000
        // tasks is 0.1 ms in front of the launch)
        // P chunks of the task list count, one thread turns the counts into places, the chunks fill: every group in ascending task order
        const size_t P = n >= 32768 ? 16 : 1;
        auto chunk = [&](size_t c) { return n * c / P; };
        auto on_chunks = [&](auto&& fn) {
            if (P == 1) { fn((size_t)0); return; }
            auto body = [&](size_t lo, size_t hi) { for (size_t c = lo; c < hi; c++) fn(c); };
            HostPool::get().run(P, body);
        };
        std::vector<size_t> cnt(P * K_COUNT, 0);
        std::vector<u64> rws(P * K_COUNT, 0);
        on_chunks([&](size_t c) {
            size_t cc[K_COUNT] = {0};   // (on the chunk's own stack: the rows of `cnt` share cache lines)
            u64 rr[K_COUNT] = {0};
            for (size_t i = chunk(c); i < chunk(c + 1); i++) if (kidv[i] >= 0) { cc[kidv[i]]++; rr[kidv[i]] += rowsv[i]; }
            for (int k = 0; k < K_COUNT; k++) { cnt[c * K_COUNT + k] = cc[k]; rws[c * K_COUNT + k] = rr[k]; }
        });
        for (int k = 0; k < K_COUNT; k++) {
            size_t run = 0;
            for (size_t c = 0; c < P; c++) { const size_t v = cnt[c * K_COUNT + k]; cnt[c * K_COUNT + k] = run; run += v; rows_of[k] += rws[c * K_COUNT + k]; }
            groups[k].resize(run);
        }
        on_chunks([&](size_t c) {
            size_t at[K_COUNT];
            for (int k = 0; k < K_COUNT; k++) at[k] = cnt[c * K_COUNT + k];
            for (size_t i = chunk(c); i < chunk(c + 1); i++) if (kidv[i] >= 0) groups[kidv[i]][at[kidv[i]]++] = (u32)i;
        });
    }

    // Small band-150 batches (merge-block rounds): one launch instead of two.  The launches of a batch run one after the
    // other on the context's stream and each lasts as long as its longest task, so a round in which some contigs hold N
    // paid for two sweeps; the N-aware kernel aligns N-free contigs too (same results, ~4 % more time per cell, which a
    // latency-bound round does not notice).  Only while everything is resident at once; GAMDP_NO_MERGE_N=1 keeps the split
    // (A/B measurements); not when GAMDP_QUAD_MIN forces the throughput kernels (tests).
    {
        static const bool keep_split = std::getenv("GAMDP_NO_MERGE_N") != nullptr || std::getenv("GAMDP_QUAD_MIN") != nullptr;
        auto &f = groups[K_C5_CE0], &a = groups[K_C5_CE0_N];
        if (!keep_split && !f.empty() && !a.empty() && f.size() + a.size() <= (size_t)n_cu * (size_t)kernel_waves_per_cu(K_C5_CE0_N)) {
            for (u32 i : f) { prep[i].kid = K_C5_CE0_N; kidv[i] = (int8_t)K_C5_CE0_N; }
            rows_of[K_C5_CE0_N] += rows_of[K_C5_CE0]; rows_of[K_C5_CE0] = 0;
            std::vector<u32> all(f.size() + a.size());
            std::merge(f.begin(), f.end(), a.begin(), a.end(), all.begin());   // both ascending: stays ascending
            a.swap(all);
            f.clear();
        }
    }

    // Band 150 has two shapes: one task per wavefront (5 columns per lane: the lowest latency per task) and four tasks
    // per wavefront (19 columns per lane, direction-free fill: ~1.7x the throughput).  A batch with more band-150 tasks
    // than the chip has wave slots is throughput-bound and takes the second; merge-block rounds of a few hundred or
    // thousand calls keep the first.  (GAMDP_QUAD_MIN overrides the threshold; results do not depend on it.)
    {
        static const long quad_min = [] { const char* e = std::getenv("GAMDP_QUAD_MIN"); return e ? std::atol(e) : -1L; }();
        const size_t thr = quad_min >= 0 ? (size_t)quad_min : (size_t)n_cu * (size_t)kernel_waves_per_cu(0);
        const int from[2] = {K_C5_CE0, K_C5_CE0_N};
        static const bool no_pair150 = std::getenv("GAMDP_NO_PAIR") != nullptr;
        for (int v = 0; v < 2; v++) {
            auto& g = groups[from[v]];
            if (g.empty() || g.size() < std::max<size_t>(thr, 1)) continue;
            // ... and long enough: below ~8 k rows the top / end blocks and the one-after-the-other walks of a
            // wavefront eat what the fill gains (measured: 400 000 x 2 kb pairs 15 % slower, 5 kb equal, 20 kb 8 % faster)
            const u64 rows = rows_of[from[v]];
            // without N and with enough tasks to fill the chip eight at a time: two quads per wavefront, packed f16 (round 3: from
            // ~4 k rows on: 400 000 x 5 kb pairs measured 5 % faster than one task per wavefront, 9 % faster than four)
            // ... and, for long contigs, from 6 144 tasks on: more than the one-task kernel holds in one round (5 120), and a
            // sparsely filled eight-task launch beats both its second round and the four-task int32 kernel (50 kb pairs:
            // 6 144 tasks 19.3 against 24.1 ms, 16 384 tasks 32.1 against 41.4 ms (four-task kernel), 4 096 tasks 18.6 against 12.3)
            // (round 4, with the top blocks packed and the strips centred on the band's middle column: from ~1.5 k rows on -- 400 000 x 2 kb
            // 4 450 -> 5 500 GCUPS, x 3 kb 4 980 -> 6 200, x 1 kb 3 300 against 3 100 the other way; GAMDP_OCTO_MIN_ROWS overrides, A/B)
            static const size_t octo_min_rows = [] { const char* e = std::getenv("GAMDP_OCTO_MIN_ROWS"); return e ? (size_t)std::min(std::max(std::atol(e), 0L), 500000L) : (size_t)0; }();
            // ... and from 12 288 calls on, not only from a chip-full of eight-task wavefronts (32 768): 16 384 x 5 kb 6.0 -> 5.0 ms, 24 576 x 5 kb
            // 8.7 -> 7.3, 16 384 x 2.5 kb 3.8 - 4.9 -> 3.7 - 3.8; 8 192 x 5 kb: equal
            // End of round 5, with the end / top / ramp blocks of a unit at half their cost (whole calls, one-task against eight-task kernel):
            //   from 16 384 calls on at every length measured (x 0.5 kb 2.34 -> 1.91 ms, x 1 kb 2.51 -> 2.19; 65 536 x 0.5 kb 7.7 -> 6.4, x 1.3 kb 10.8 -> 8.0),
            //   from 12 288 calls of >= 1 k rows (x 1 kb: equal), from 8 192 of >= 2.5 k (x 3 kb 2.53 -> 2.19, x 5 kb 3.73 -> 2.79; x 2 kb: equal),
            //   from 6 144 of >= 4.5 k (x 5 kb 2.73 -> 2.55; x 3 kb: equal); 4 096 calls: the one-task kernel at every length.
            // (GAMDP_OCTO_MIN_ROWS=r: at least r rows on average whatever the count -- A/B, and the tests' way to the other kernel)
            const size_t avg_rows = rows / g.size();
            auto tier = [&](size_t calls, size_t min_rows) { return g.size() >= calls && avg_rows >= std::max(min_rows, octo_min_rows); };
            const bool octo = v == 0 && !no_pair150 && !diag_no_dirfree &&
                              (quad_min >= 0 ? g.size() >= (size_t)quad_min : (tier(16384, 384) || tier(12288, 1024) || tier(8192, 2560) || tier(6144, 4608)));
            if (!octo && quad_min < 0 && rows / g.size() < 8192) continue;
            const int to = octo ? K_O19_CE15 : (v == 0 ? K_Q19_CE15 : K_Q19_CE15_N);
            // (nothing to rewrite per task: the planner below sizes the slots of a one-band kernel from its longest task)
            groups[to] = std::move(g);
            g.clear();
        }
    }

    // Band 512 without N: two tasks per wavefront, their common fast blocks in packed f16 (kernel_pair.inc) -- whenever
    // there are two tasks to pair.  (GAMDP_NO_PAIR=1 keeps the one-task kernel: A/B measurements; results do not
    // depend on it.  The diagnostics switch that keeps a direction per cell also does.)
    {
        static const bool no_pair = std::getenv("GAMDP_NO_PAIR") != nullptr;
        auto& g = groups[K_C17_CE4];
        if (!no_pair && !diag_no_dirfree && g.size() >= 2) {
            groups[K_P17_CE4] = std::move(g);
            g.clear();
        }
    }

    const double ms_prep = since(t_begin);
    mark("groups");
    int rc_ = grow(this, d_results, cap_results, n + 1);  // + one dump slot for the padding tasks of the 4-task kernels
    if (rc_) return rc_;
    if (ops_total) { rc_ = grow(this, d_ops, cap_ops, ops_total); if (rc_) return rc_; }

    // arena budget of THIS call: the context's budget over the contexts that share the device right now
    if (arena_budget() == 0) { set_error("hipMemGetInfo failed"); return GAMDP_EHIP; }
    const u64 arena_call = this->arena_call();

    u64 n_host_tasks = 0;
    std::vector<u64>& cells_key = w_key;
    struct Launch { int kid; u32 first, count; u64 slot_words, dir_words; u32 ypad, n_slots, dyn_lds; u64 ckpt_off, bnd_off; u32 band_max; u64 scratch_off = 0; bool aux = false; };
    std::vector<Launch> launches;
    std::vector<std::vector<u32>> launch_items;   // the tasks of every launch, in launch order (staged once the plan is complete)
    // (Measured and dropped: handing the leftover of a multi-task group -- less than one round -- to a finer-grained kernel
    // as a launch of its own, and starting every other wavefront half a fill late to take the wavefronts of a launch of
    // equal tasks out of lock-step.  The first costs more than the stragglers do (launches are sequential), the second
    // changed nothing beyond noise.)
    for (int kid = 0; kid < K_COUNT; kid++) {
        auto& g = groups[kid];
        if (g.empty()) continue;
        const u32 max_resident = (u32)n_cu * (u32)kernel_waves_per_cu(kid);
        // longest tasks first (LPT); ties keep the caller's order.  The tuned kernels take one band each: their tasks' cells are in the order
        // of their rows, a key of 19 bits at most instead of 40 (two passes of the radix sort instead of three)
        if (kid <= K_Q19_CE15_N) sort_by_key_desc(g, rowsv, &w_sort_tmp, &w_sort_count);
        else sort_by_key_desc(g, cells_key, &w_sort_tmp, &w_sort_count);
        // One launch per group if slots sized for its largest direction matrix leave enough resident
        // waves; otherwise peel off the tasks with big matrices into their own launch and retry.
        std::vector<std::vector<u32>> work;
        work.push_back(g);   // (a copy: the group's own list keeps its memory for the next call)
        while (!work.empty()) {
            std::vector<u32> cur = std::move(work.back());
            work.pop_back();
            u32 maxband = kernel_band(kid);
            u64 maxdir = 0;
            if (maxband != 0) maxdir = dir_words_for(kid, rowsv[cur[0]], maxband);   // a one-band kernel: the list is sorted by rows, the longest task first
            else {   // (a reduction over the descriptors in sorted -- i.e. random -- order: in parallel for big launches)
                std::mutex red;
                parallel_for(cur.size(), [&](size_t lo, size_t hi) {
                    u32 mb = 0; u64 md = 0;
                    for (size_t k = lo; k < hi; k++) { const u32 i = cur[k]; mb = std::max<u32>(mb, (u32)prep[i].dt.band); md = std::max(md, prep[i].dir_words); }
                    std::lock_guard<std::mutex> gd(red);
                    maxband = std::max(maxband, mb); maxdir = std::max(maxdir, md);
                });
            }
            const u32 one_band = kernel_band(kid);
            const u32 ypad = ((2 * maxband + 2 + 63) / 64) * 64;
            const u64 dirw = ((maxdir + 63) / 64) * 64;
            // the tuned N-free kernels fill their fast blocks without directions and keep, per 4 blocks, one live row
            // (C*64 words) and, per block, 512 boundary words instead (gamdp_kernel.hip, do_block_df)
            u64 ckpt_words = 0, bnd_words = 0;
            const u32 tpw = (u32)kernel_tasks_per_wave(kid);  // tasks per wavefront: each has its own side buffers
            const bool pair = kid == K_P17_CE4 || kid == K_O19_CE15;   // ... and, for the pairs (of tasks / of quads), its own direction words
            if (tpw > 1 || kernel_dirfree(kid)) {   // (the tuned band-512 kernels, the multi-task kernels, the generic kernels with 9 / 17 columns per lane)
                const u64 cw = (u64)kernel_dir_block_words(kid), nblk = dirw / cw + 1;
                ckpt_words = (nblk / 4 + 2) * (u64)kernel_ckpt_words(kid);
                bnd_words = (nblk + 4) * (u64)kernel_bnd_words(kid);
            }
            const u64 dir_total = pair ? 2 * dirw : dirw;
            const u64 slotw = dir_total + 4ull * ypad * tpw + ckpt_words + bnd_words;
            const u64 fit = arena_call / (slotw * sizeof(u32));
            if (fit == 0) { set_error("scratch arena too small for one task"); return GAMDP_ENOMEM; }
            const u64 want = std::min<u64>((cur.size() + tpw - 1) / tpw, max_resident);
            if (fit < want && cur.size() > 1) {
                std::vector<u32> big, small;
                for (u32 i : cur) ((one_band ? dir_words_for(kid, rowsv[i], one_band) : prep[i].dir_words) * 2 >= maxdir ? big : small).push_back(i);
                if (!small.empty()) {
                    work.push_back(std::move(small));
                    work.push_back(std::move(big));  // processed first (largest tasks first)
                    continue;
                }
            }
            Launch L;
            const size_t padded = (cur.size() + tpw - 1) / tpw * tpw;
            L.kid = kid; L.first = (u32)n_host_tasks; L.count = (u32)padded;
            L.slot_words = slotw; L.dir_words = dirw; L.ypad = ypad; L.band_max = maxband;
            // (Measured and dropped: equalising the rounds of a launch -- 6 250 workgroups as 2 x 3 125 instead of 4 096 +
            // 2 154 -- and forcing an even spread over the CUs with unused dynamic LDS changed nothing: the hardware
            // dispatcher already spreads workgroups evenly, and what a short launch loses is per-SIMD occupancy.)
            L.n_slots = (u32)std::min<u64>(want, fit);
            L.dyn_lds = 0;
            L.ckpt_off = ckpt_words ? dir_total + 4ull * ypad * tpw : 0;
            L.bnd_off = L.ckpt_off + ckpt_words;
            n_host_tasks += padded;
            launch_items.push_back(std::move(cur));
            launches.push_back(L);
        }
    }
    // A small launch beside a big one (round 6).  The launches of a call run one after the other and each lasts as long as its longest
    // task: the ~900 calls of a driver-shaped batch whose windows hold N -- one one-task wavefront each, less than one per SIMD -- kept
    // the chip for 1.35 ms before the eight-task launch of the other 99 000 began (11.6 % of the call's kernel time for 0.9 % of its
    // calls).  Such a launch (one round of a one-task kernel, at most one wavefront per SIMD) now goes FIRST on a stream of its own, with
    // its own piece of the task upload and its own region of the scratch arena, and the big launch (two rounds or more of a multi-task
    // kernel) starts beside it: the small launch's wavefronts are the oldest on their SIMDs, which at equal priority get every issue
    // slot they can use (kernel_common.inc), and three eight-task wavefronts per SIMD keep the vector pipe busy meanwhile.
    // Results cannot depend on it: same kernels, same tasks, disjoint scratch.  GAMDP_NO_AUX_LAUNCH=1: one after the other (A/B).
    {
        static const bool no_aux = std::getenv("GAMDP_NO_AUX_LAUNCH") != nullptr;
        int small = -1, big = -1;
        for (size_t li = 0; li < launches.size(); li++) {
            const Launch& L = launches[li];
            const u64 tpw = (u64)kernel_tasks_per_wave(L.kid), units = L.count / tpw;
            if (tpw == 1 && L.kid != K_WIDE && units <= (u64)L.n_slots && units <= 4ull * (u64)n_cu) { if (small < 0) small = (int)li; }
            else if (tpw > 1 && units >= 2ull * L.n_slots) { if (big < 0) big = (int)li; }
        }
        if (!no_aux && !interval_sink && !defer_frees && small >= 0 && big >= 0 && launches.size() == 2) {
            const u64 main_words = launches[big].slot_words * launches[big].n_slots, small_words = launches[small].slot_words * launches[small].n_slots;
            if ((main_words + small_words) * sizeof(u32) <= arena_call) {   // (both regions inside this call's share of the arena)
                launches[small].aux = true;
                launches[small].scratch_off = main_words;
            }
        }
    }
    mark("sort+plan");
    // Staging: the padded task list of all launches (the last wavefront of a multi-task launch is filled up with copies of
    // its last task that write to the dump slot), sized from the finished plan -- however many launches the peeling made.
    rc_ = grow(this, d_tasks, cap_tasks, n_host_tasks + 1);
    if (rc_) return rc_;
    // pinned staging for the task upload and the result download (pageable copies cost ~20 ms per 60 k tasks)
    if (std::max<u64>(n_host_tasks, n) + 1 > cap_pinned) {
        free_host(h_tasks);
        free_host(h_results);
        h_tasks = nullptr; h_results = nullptr; cap_pinned = 0;
        const u64 base = std::max<u64>(n_host_tasks, n), want = base + base / 4 + 256;
        if (hipHostMalloc(&h_tasks, want * sizeof(DevTask)) != hipSuccess ||
            hipHostMalloc(&h_results, want * sizeof(DevResult)) != hipSuccess) {
            set_error("hipHostMalloc of staging buffers failed");
            return GAMDP_ENOMEM;
        }
        cap_pinned = want;
    }
    for (size_t li = 0; li < launches.size(); li++) {
        const std::vector<u32>& cur = launch_items[li];
        const Launch& L = launches[li];
        DevTask* dst = h_tasks + L.first;
        parallel_for(cur.size(), [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) dst[k] = prep[cur[k]].dt; });
        // the last wavefront of a multi-task launch: filled up with copies of its last call that are filled along and nothing else
        auto pad = [&](size_t k, const DevTask& of) { dst[k] = of; dst[k].res_idx = (u32)n; dst[k].flags = (dst[k].flags & ~(u32)TF_WANT_OPS) | TF_PADDING; };
        size_t k = cur.size();
        for (; k < L.count; k++) pad(k, dst[cur.size() - 1]);
    }
    mark("staging");
    if (!launches.empty()) {
        u64 need_scratch = 0;
        for (auto& L : launches) need_scratch = std::max(need_scratch, L.scratch_off + L.slot_words * L.n_slots);
        if (need_scratch > cap_scratch) {
            if (d_scratch) { free_dev(d_scratch); d_scratch = nullptr; cap_scratch = 0; }
            // a round loop beside a chain launch: what a round holds depends on which chains have ended, so its needs move from
            // call to call -- room to spare (inside the call's share) instead of a hipMalloc every few calls (1.5 ms each)
            if (defer_frees) need_scratch = std::max(need_scratch, std::min<u64>(2 * need_scratch, arena_call / sizeof(u32)));
            if (hipMalloc(&d_scratch, need_scratch * sizeof(u32)) != hipSuccess) {
                // a call on this context alone after calls that shared the device: its idle helper contexts may still
                // hold their shares of the budget
                d_scratch = nullptr;
                (void)hipGetLastError();
                flush_frees();   // (buffers kept back while a chain launch runs: now they have to go, whatever the wait)
                if (arena_div == 1) {
                    for (Ctx* h : helpers)
                        if (h->d_scratch) { (void)hipFree(h->d_scratch); h->d_scratch = nullptr; h->cap_scratch = 0; }
                    if (d_chain_scratch) { (void)hipFree(d_chain_scratch); d_chain_scratch = nullptr; cap_chain_scratch = 0; }   // (idle: merge-block calls are synchronous)
                }
                if (hipMalloc(&d_scratch, need_scratch * sizeof(u32)) != hipSuccess) {
                    d_scratch = nullptr;
                    set_error("hipMalloc of scratch arena (" + std::to_string(need_scratch * sizeof(u32)) + " bytes) failed");
                    return GAMDP_ENOMEM;
                }
            }
            cap_scratch = need_scratch;
        }
        if (launches.size() > cap_cursor) {
            // cursors: one u32 per launch, and LS_COUNT statistics words per launch behind them
            free_dev(d_cursor);
            d_cursor = nullptr; cap_cursor = 0;
            HIPCHK(this, hipMalloc(&d_cursor, launches.size() * (1 + LS_COUNT) * sizeof(u32)));
            cap_cursor = (u32)launches.size();
        }
        u32* const d_stats = d_cursor + cap_cursor;
        const double ms_plan = since(t_begin) - ms_prep;
        mark("buffers");
        const auto t_gpu = std::chrono::steady_clock::now();
        bool any_aux = false;
        for (auto& L : launches) any_aux |= L.aux;
        HIPCHK(this, hipMemsetAsync(d_cursor, 0, (size_t)cap_cursor * (1 + LS_COUNT) * sizeof(u32), stream));
        if (any_aux) {
            if (!aux_stream) {
                HIPCHK(this, hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking));
                HIPCHK(this, hipEventCreateWithFlags(&aux_done, hipEventDisableTiming));
                HIPCHK(this, hipEventCreateWithFlags(&aux_go, hipEventDisableTiming));
            }
            // the cursors are zero before anybody starts; the small launch's tasks go up on its own stream, ahead of the big upload
            HIPCHK(this, hipEventRecord(aux_go, stream));
            HIPCHK(this, hipStreamWaitEvent(aux_stream, aux_go, 0));
            for (auto& L : launches)
                if (L.aux) HIPCHK(this, hipMemcpyAsync(d_tasks + L.first, h_tasks + L.first, (size_t)L.count * sizeof(DevTask), hipMemcpyHostToDevice, aux_stream));
            for (auto& L : launches)
                if (!L.aux) HIPCHK(this, hipMemcpyAsync(d_tasks + L.first, h_tasks + L.first, (size_t)L.count * sizeof(DevTask), hipMemcpyHostToDevice, stream));
        } else {
            HIPCHK(this, hipMemcpyAsync(d_tasks, h_tasks, n_host_tasks * sizeof(DevTask), hipMemcpyHostToDevice, stream));
        }
        while (events.size() < launches.size()) {
            hipEvent_t a, b;
            HIPCHK(this, hipEventCreate(&a));
            HIPCHK(this, hipEventCreate(&b));
            events.push_back({a, b});
        }
        for (size_t li = 0; li < launches.size(); li++) {
            const Launch& L = launches[li];
            LaunchParams p;
            p.tasks = d_tasks + L.first; p.n_tasks = L.count; p.cursor = d_cursor + li;
            p.results = d_results; p.ops_buf = d_ops;
            p.scratch = d_scratch + L.scratch_off; p.slot_words = L.slot_words; p.dir_words = L.dir_words; p.ypad = L.ypad;
            p.ckpt_off = L.ckpt_off; p.bnd_off = L.bnd_off;
            p.stats = d_stats + li * LS_COUNT;
            static const u64 side_rounds = [] { const char* e = std::getenv("GAMDP_SIDE_WALK_ROUNDS"); return e ? (u64)std::min(std::max(std::atol(e), 0L), 1000000L) : 2ull; }();
            p.flags = (L.kid == K_P17_CE4 && (u64)L.count / 2 <= side_rounds * L.n_slots) ? LP_WALK_SIDE_BY_SIDE : 0u;
            static const bool no_packed_top = std::getenv("GAMDP_NO_PACKED_TOP") != nullptr;
            if (no_packed_top) p.flags |= LP_NO_PACKED_TOP;
            static const bool no_packed_top_mixed = std::getenv("GAMDP_NO_PACKED_TOP_MIXED") != nullptr;
            if (no_packed_top_mixed) p.flags |= LP_NO_PACKED_TOP_MIXED;
            static const bool no_strip_shift = std::getenv("GAMDP_NO_STRIP_SHIFT") != nullptr;
            if (no_strip_shift) p.flags |= LP_NO_STRIP_SHIFT;
            {   // the end-cell / strip / walk phase of the two- and eight-task kernels issues one level above a steady-state fill in
                // launches of more than two rounds (gamdp_dev.h; GAMDP_WALK_PRIO=0..3 overrides, A/B)
                static const int walk_prio = [] { const char* e = std::getenv("GAMDP_WALK_PRIO"); return e ? std::atoi(e) & 3 : -1; }();
                const u64 units = L.count / (u64)kernel_tasks_per_wave(L.kid);
                p.flags |= (walk_prio >= 0 ? (u32)walk_prio : (units > 2ull * L.n_slots ? 1u : 0u)) << LP_WALK_PRIO_SHIFT;
            }
            {   // longest-remaining-first issue priority for the units in flight when the queue runs dry (gamdp_dev.h)
                static const bool no_prio = std::getenv("GAMDP_NO_PRIO") != nullptr;
                const u64 tpw = (u64)kernel_tasks_per_wave(L.kid), units = L.count / tpw;
                p.prio_R = no_prio ? 0u : (u32)std::max<u64>(1, L.dir_words / (u64)kernel_dir_block_words(L.kid));
                // the units of the last round -- of the last two for the two-task kernel (measured: band 512 352.6 -> 349.9 ms; the eight-task
                // kernel loses 1 % with two, 4 % with all)
                const u64 shaped = (L.kid == K_P17_CE4 ? 2ull : 1ull) * L.n_slots;
                p.prio_from = units <= 2ull * L.n_slots ? 0u : (u32)(units - std::min<u64>(units, shaped));
            }
            hipStream_t const on = L.aux ? aux_stream : stream;
            HIPCHK(this, hipEventRecord(events[li].first, on));
            const int e = launch_align(L.kid, p, L.n_slots, L.dyn_lds, on);
            if (e != 0) { set_error(std::string("kernel launch failed: ") + hipGetErrorString((hipError_t)e)); return GAMDP_EHIP; }
            HIPCHK(this, hipEventRecord(events[li].second, on));
        }
        if (any_aux) {   // the downloads below wait for both
            HIPCHK(this, hipEventRecord(aux_done, aux_stream));
            HIPCHK(this, hipStreamWaitEvent(stream, aux_done, 0));
        }
        DevResult* hres = h_results;
        HIPCHK(this, hipMemcpyAsync(hres, d_results, n * sizeof(DevResult), hipMemcpyDeviceToHost, stream));
        std::vector<uint8_t> hops(ops_total);
        if (ops_total) HIPCHK(this, hipMemcpyAsync(hops.data(), d_ops, ops_total, hipMemcpyDeviceToHost, stream));
        std::vector<u32> hstats(log_launches ? launches.size() * LS_COUNT : 0);
        if (!hstats.empty()) HIPCHK(this, hipMemcpyAsync(hstats.data(), d_stats, hstats.size() * sizeof(u32), hipMemcpyDeviceToHost, stream));
        mark("enqueued");
        HIPCHK(this, hipStreamSynchronize(stream));
        mark("gpu done");
        const double ms_gpu = since(t_gpu);
        float busy_hi = 0;   // (side-by-side launches: the end of the busy time so far, ms after the first launch began)
        for (size_t li = 0; li < launches.size(); li++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, events[li].first, events[li].second) == hipSuccess) {
                kernel_launches++;
                if (!any_aux) kernel_ms += ms;
                else {
                    // launches that ran side by side: the context's kernel time is the time the GPU was busy with them (the union of
                    // their intervals on the device's clock), not the sum of their durations
                    float off = 0;
                    if (li > 0 && hipEventElapsedTime(&off, events[0].first, events[li].first) != hipSuccess) off = busy_hi;
                    const float lo = std::max(off, busy_hi), hi = off + ms;
                    if (hi > lo) kernel_ms += hi - lo;
                    busy_hi = std::max(busy_hi, hi);
                }
            }
            if (log_launches) {   // gamdp_ctx_launch_info: the planner's choice and what the device counted
                const Launch& L = launches[li];
                gamdp_launch_info r;
                std::memset(&r, 0, sizeof(r));
                std::snprintf(r.kernel, sizeof(r.kernel), "%s", kernel_name(L.kid));
                r.n_aware = kernel_n_aware(L.kid) ? 1u : 0u;
                r.tasks_per_wavefront = (u32)kernel_tasks_per_wave(L.kid);
                r.tasks = (u32)launch_items[li].size();
                r.units = L.count / r.tasks_per_wavefront;
                r.slots = L.n_slots;
                r.band_max = L.band_max;   // (from the planner's reduction: a pass over the launch's descriptors here was 1 ms per 100 000 tasks)
                const u32* st = hstats.data() + li * LS_COUNT;
                r.units_dirfree = st[LS_DIRFREE]; r.units_packed_top = st[LS_PACKED_TOP]; r.units_packed_top_mixed = st[LS_PACKED_TOP_MIXED];
                r.strips = st[LS_STRIPS]; r.units_top_wanted = st[LS_TOP_WANTED];
                r.piece = log_piece;
                r.rounds = L.n_slots ? (double)r.units / (double)L.n_slots : 0.0;
                r.kernel_ms = (double)ms;
                launch_log.push_back(r);
            }
            if (interval_sink && ref_event) {  // merge-block calls: where this launch sat on the call's time line
                float t0 = 0;
                if (hipEventElapsedTime(&t0, ref_event, events[li].first) == hipSuccess) interval_sink->push_back({t0, t0 + ms});
            }
        }
        const auto t_fill = std::chrono::steady_clock::now();
        mark("events");
        // results by task index (round 6: both arrays in order -- by launch position it was a random read and a random write per task)
        parallel_for(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                if (kidv[i] < 0) continue;   // settled by the pre-checks
                fill_result(hres[i], cells_key[i], out[i]);   // (cells_key[i] = prep[i].cells, 8 bytes apart instead of 120)
            }
        });
        if (ops_total) {
            parallel_for(n_host_tasks, [&](size_t lo, size_t hi) {
                for (size_t q = lo; q < hi; q++) {
                    const DevTask& d = h_tasks[q];
                    const u32 i = d.res_idx;
                    if (i >= n) continue;  // padding of a multi-task launch
                    if ((d.flags & TF_WANT_OPS) && out[i].status == GAMDP_ST_OK) {
                        const u64 len = std::min<u64>(out[i].length, d.ops_cap);
                        // the kernel wrote ops in traceback order: reverse into the caller's buffer
                        uint8_t* dst = ops->ops_buf + ops->ops_off[i];
                        const uint8_t* src = hops.data() + d.ops_off;
                        if (out[i].length <= d.ops_cap) for (u64 k = 0; k < len; k++) dst[k] = src[len - 1 - k];
                        else for (u64 k = 0; k < len; k++) dst[k] = 0xFF;  // truncated: not reconstructible
                    }
                }
            });
        }
        if (diag_timing) {
            mark("results");
            // (every line at the END of the call: a write to stderr costs 0.2 - 2 ms on the GPU boxes, and in the middle of the call it
            // was the largest item of the "results" phase it reported.  ctx + begin on the process's steady clock: tools/multi_host_overlap.py
            // lays the host phases of several contexts side by side)
            std::fprintf(stderr, "libgamdp align: %zu tasks, %zu launches: prepare %.2f ms, plan+stage %.2f ms, upload+kernels+download %.2f ms [ctx %p began %.3f]\n",
                         n, launches.size(), ms_prep, ms_plan, ms_gpu, (void*)this,
                         std::chrono::duration<double, std::milli>(t_begin.time_since_epoch()).count());
            std::fprintf(stderr, "libgamdp align: results %.2f ms [ctx %p began %.3f]\n", marks.back().second - std::chrono::duration<double, std::milli>(t_fill - t_begin).count(), (void*)this,
                         std::chrono::duration<double, std::milli>(t_fill.time_since_epoch()).count());
            std::string line = "libgamdp align: phases (ms since the call began):";
            for (auto& m : marks) { char b[64]; std::snprintf(b, sizeof b, " %s %.2f", m.first, m.second); line += b; }
            std::fprintf(stderr, "%s; kernels %.2f ms\n", line.c_str(), [&] { double k = 0; for (auto& r : launch_log) k += r.kernel_ms; return k; }());
        }
    }
    return 0;
}

}  // namespace gamdp

// ---- C ABI ----------------------------------------------------------------------------------------
using namespace gamdp;

extern "C" {

int gamdp_ctx_create(int device, gamdp_ctx** out)
{
    if (!out) return GAMDP_EINVAL;
    *out = nullptr;
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return GAMDP_ENOMEM;
    const int rc_ = c->init(device);
    if (rc_) {
        std::fprintf(stderr, "libgamdp: %s\n", c->err.c_str());
        delete c;
        return rc_;
    }
    *out = reinterpret_cast<gamdp_ctx*>(c);
    return 0;
}

void gamdp_ctx_destroy(gamdp_ctx* ctx) { delete reinterpret_cast<Ctx*>(ctx); }

int gamdp_ctx_set_arena_bytes(gamdp_ctx* ctx, uint64_t bytes)
{
    if (!ctx) return GAMDP_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    c->arena_limit = bytes;                           // 0 = back to the automatic budget
    c->arena_auto = bytes == 0;
    for (Ctx* h : c->helpers) h->arena_limit = bytes; // (every call hands the owner's budget to its helpers again anyway)
    return 0;
}

const char* gamdp_last_error(const gamdp_ctx* ctx) { return ctx ? reinterpret_cast<const Ctx*>(ctx)->err.c_str() : "null ctx"; }

void* gamdp_ctx_stream(gamdp_ctx* ctx) { return ctx ? (void*)reinterpret_cast<Ctx*>(ctx)->stream : nullptr; }

int gamdp_ctx_kernel_time(gamdp_ctx* ctx, double* total_ms, uint64_t* launches, int reset)
{
    if (!ctx) return GAMDP_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (total_ms) *total_ms = c->kernel_ms;
    if (launches) *launches = c->kernel_launches;
    if (reset) { c->kernel_ms = 0; c->kernel_launches = 0; }
    return 0;
}

int gamdp_ctx_launch_info(const gamdp_ctx* ctx, gamdp_launch_info* out, size_t cap, size_t* n)
{
    if (!ctx || (cap && !out)) return GAMDP_EINVAL;
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    if (n) *n = c->launch_log.size();
    for (size_t i = 0; i < c->launch_log.size() && i < cap; i++) out[i] = c->launch_log[i];
    return 0;
}

int gamdp_seqset_create(gamdp_ctx* ctx, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, int is_ascii,
                        gamdp_seqset** out)
{
    if (!ctx || !out || (n && (!seqs || !lens))) return GAMDP_EINVAL;
    *out = nullptr;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (hipSetDevice(c->device) != hipSuccess) return GAMDP_EHIP;
    SeqSet* s = new (std::nothrow) SeqSet();
    if (!s) return GAMDP_ENOMEM;
    const int rc_ = guarded(c, [&] { return s->upload(c, seqs, lens, n, is_ascii != 0); });
    if (rc_) { delete s; return rc_; }
    *out = reinterpret_cast<gamdp_seqset*>(s);
    return 0;
}

void gamdp_seqset_destroy(gamdp_seqset* set)
{
    SeqSet* s = reinterpret_cast<SeqSet*>(set);
    if (s && s->ctx) (void)hipSetDevice(s->ctx->device);
    delete s;
}

uint32_t gamdp_seqset_size(const gamdp_seqset* set) { return set ? (uint32_t)reinterpret_cast<const SeqSet*>(set)->lens.size() : 0; }

uint64_t gamdp_seqset_length(const gamdp_seqset* set, uint32_t id)
{
    const SeqSet* s = reinterpret_cast<const SeqSet*>(set);
    return (s && id < s->lens.size()) ? s->lens[id] : 0;
}

int gamdp_align_batch(gamdp_ctx* ctx, const gamdp_seqset* set_a, const gamdp_seqset* set_b, const gamdp_task* tasks,
                      size_t n, gamdp_result* out, const gamdp_ops* ops)
{
    if (!ctx || !set_a || !set_b || (n && (!tasks || !out))) return GAMDP_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const SeqSet* sa = reinterpret_cast<const SeqSet*>(set_a);
    const SeqSet* sb = reinterpret_cast<const SeqSet*>(set_b);
    c->launch_log.clear();
    struct LogGuard { Ctx* c; ~LogGuard() { c->log_launches = false; for (Ctx* h : c->helpers) h->log_launches = false; } } log_guard{c};
    c->log_launches = true; c->log_piece = 0;
    return guarded(c, [&]() -> int {
    const auto t_call = std::chrono::steady_clock::now();
    if (c->arena_budget(true) == 0) { c->set_error("hipMemGetInfo failed"); return GAMDP_EHIP; }
    const double ms_arena = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    struct CallTimer { std::chrono::steady_clock::time_point t0; size_t n; const double* arena; ~CallTimer() { if (diag().timing) std::fprintf(stderr, "libgamdp align_batch: %zu calls, %.2f ms in all (arena budget %.2f ms)\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), *arena); } } call_timer{t_call, n, &ms_arena};
    auto run = [&](Ctx* cc, size_t first, size_t cnt) -> int {
        TaskSrc src;   // the caller's array, read in place
        src.gt = tasks + first; src.sa = sa; src.sb = sb;
        return cc->align(src, cnt, out + first, nullptr);
    };
    // Very large batches of small calls (hundreds of thousands): validation, sorting, staging and result conversion of the
    // whole batch would sit in front of / behind the kernel (25 ms of a 120 ms step for 400 000 5 kb pairs).  They go
    // through in four pieces on two host threads with a context (stream, staging, arena) each: one piece's host work
    // runs while the other's kernel does.  Results do not depend on the split.
    // (From 65 536 calls on when the calls are small -- by their windows under ~8 M cell updates each, 4.5 M at band 150 whose
    // eight-task wavefronts hold eight direction images per scratch slot: 11 kb at band 512, 23 kb at band 150 -- since the
    // first piece is small (below): 200 000 x 5 kb at band 512 9 110 -> 9 780 GCUPS.  Batches of long calls stay whole whatever
    // their number: the two contexts share the scratch arena, and four short launches would lose more
    // at their ends than the host work they hide (100 000 x 50 kb in pieces: 2.5 s per step).  GAMDP_CHUNK_MIN=n: from n calls
    // on, whatever their size.)
    static const long chunk_env = [] { const char* e = std::getenv("GAMDP_CHUNK_MIN"); return e ? (long)std::atoll(e) : -1L; }();
    bool chunked = chunk_env >= 0 && n >= (size_t)chunk_env;
    bool b150 = false;   // (the band of most of the sampled calls)
    if (chunked) { size_t n150 = 0, cnt = 0; for (size_t i = 0; i < n; i += 64, cnt++) n150 += tasks[i].band == 150; b150 = 2 * n150 >= cnt; }
    if (chunk_env < 0 && n >= 65536) {
        double est = 0;   // (a sample of the windows is enough: every 64th call)
        size_t cnt = 0;
        size_t n150 = 0;
        u64 rows_max = 0, rows_sum = 0;
        for (size_t i = 0; i < n; i += 64, cnt++) {
            const gamdp_task& t = tasks[i];
            // rows as the pre-checks will size them (banded_smith_waterman.cc:91-95): end_b clipped to the contig, no wrap of the + 1
            const u64 blen = t.b_id < sb->lens.size() ? sb->lens[t.b_id] - std::min<u64>(t.b_off, sb->lens[t.b_id]) : 0;
            const u64 eb = blen ? std::min<u64>(t.end_b, blen - 1) : 0;
            const u64 rows = (blen && eb >= t.begin_b) ? std::min<u64>(eb - t.begin_b + 1, 500000) : 0;
            est += (double)rows * (2.0 * t.band + 1.0);
            rows_max = std::max(rows_max, rows); rows_sum += rows;
            n150 += t.band == 150;
        }
        // (re-measured with the walk phase's priority in place, which only launches of more than two rounds get: 100 000 x 20 kb at band 150
        // 81 ms in pieces, 72 - 79 whole; 100 000 x 10 kb at band 512 92.5 in pieces, 88.4 whole; 200 000 x 5 kb at band 512 97 - 105 in pieces,
        // 106 - 107 whole; 200 000 x 10 kb at band 150 a tie)
        b150 = 2 * n150 >= cnt;
        chunked = est / (double)std::max<size_t>(1, cnt) < (b150 ? 4.5e6 : 8e6);
        // ... and only if a piece still keeps the chip busy for a few rounds: a piece of a round or less lasts as long as its longest call,
        // four times over (the driver-shaped batch of 100 000 band-150 calls of 0.2 - 10 k rows: 26 - 29 ms in pieces of one round, 7 - 10 ms
        // each, against 16.7 ms in one LPT-balanced launch of three rounds).  Unless the calls are of one length (the longest of the sample
        // within a quarter of the mean): then pieces of whole rounds (below) lose nothing, and half a round per piece is enough -- 200 000 x 2 kb
        // 31.1 -> 20.9 ms, 262 144 x 1 kb 33.7 -> 19.9, 131 072 x 2 kb 17.8 -> 14.9, x 5 kb 29.2 -> 26.5, 65 536 x 5 kb at band 512 37.4 -> 35.1.
        // (GAMDP_CHUNK_MIN_ROUNDS: the bar for both, A/B.)
        const double units_per_piece = (double)n * (7.0 / 24.0) / (b150 ? 8.0 : 2.0);
        const bool one_length = (double)rows_max * (double)std::max<size_t>(1, cnt) <= 1.25 * (double)rows_sum;
        static const double min_rounds_env = [] { const char* e = std::getenv("GAMDP_CHUNK_MIN_ROUNDS"); return e ? std::min(std::max(std::atof(e), 0.0), 1000.0) : -1.0; }();
        // (round 6, with the host side of a call at a third of its cost -- ~1.5 ms per 100 000 calls in front of the launch: pieces pay from
        // three rounds per piece on whatever the length, and only while the host's share is worth hiding, 5 % of the kernels' time or more:
        // 400 000 x 2 kb 34.9 ms in pieces / 36.8 whole, x 5 kb 71.7 / 76.2; 300 000 x 3 kb and 200 000 x 2 kb ties; 131 072 x 2 kb 14.2 / 13.2,
        // 262 144 x 1 kb 18.6 / 17.7, 131 072 x 5 kb and 200 000 x 5 kb whole; 200 000 x 5 kb at band 512 105.0 / 101.6 (host 3 % of its
        // kernels); tools/ab_chunks_r06.sh, tools/sweep_r06.sh)
        const double min_rounds = min_rounds_env >= 0 ? min_rounds_env : (one_length ? 3.0 : 2.5);
        if (units_per_piece < min_rounds * 16.0 * (double)c->n_cu) chunked = false;
        {
            const double cells_total = est / (double)std::max<size_t>(1, cnt) * (double)n;
            const double kernel_ms = cells_total / (b150 ? 7.0e9 : 12.0e9), host_ms = (double)n * 1.5e-5;   // (rates of the packed kernels on short calls)
            if (min_rounds_env < 0 && host_ms < 0.05 * kernel_ms) chunked = false;
        }
    }
    if (!chunked || (ops && ops->ops_buf) || n < 8) {
        if (ops && ops->ops_buf) {  // edit strings (tests): the single-piece path with the caller's ops descriptor
            TaskSrc src;
            src.gt = tasks; src.sa = sa; src.sb = sb;
            return c->align(src, n, out, ops);
        }
        return run(c, 0, n);
    }
    if (hipSetDevice(c->device) != hipSuccess) { c->set_error("hipSetDevice failed"); return GAMDP_EHIP; }
    if (c->helpers.empty()) {
        Ctx* h = new (std::nothrow) Ctx();
        if (!h || h->init(c->device) != 0) { c->set_error("helper context: " + (h ? h->err : std::string("out of memory"))); delete h; return GAMDP_ENODEV; }
        c->helpers.push_back(h);
    }
    Ctx* cc[2] = {c, c->helpers[0]};
    // the two contexts share the device for this call: half the owner's budget each (the budget itself stays as it is)
    if (c->arena_budget() == 0) { c->set_error("hipMemGetInfo failed"); return GAMDP_EHIP; }
    cc[1]->arena_limit = c->arena_limit; cc[1]->arena_share = c->arena_share;
    struct DivGuard { Ctx* a; Ctx* b; ~DivGuard() { a->arena_div = 1; b->arena_div = 1; } } div_guard{cc[0], cc[1]};
    cc[0]->arena_div = cc[1]->arena_div = 2;
    cc[0]->trim_scratch(); cc[1]->trim_scratch();
    cc[1]->kernel_ms = 0; cc[1]->kernel_launches = 0;
    cc[1]->launch_log.clear(); cc[1]->log_launches = true;
    // Four pieces, the first one small: nothing runs on the GPU until the first piece is validated, sorted and staged, and
    // that costs ~5 ms per 100 000 tasks -- so the first piece is an eighth of the batch (GAMDP_CHUNK_FIRST_DIV), the other three
    // share the rest; the second thread prepares piece 1 meanwhile.
    static const size_t first_div = [] { const char* e = std::getenv("GAMDP_CHUNK_FIRST_DIV"); const long v = e ? std::atol(e) : 8; return (size_t)std::min(std::max(4L, v), 1024L); }();
    // Pieces end on whole rounds (a round = one call set per resident wavefront: 16 per CU, eight calls each at band 150, two otherwise):
    // calls of one length finish together, and a piece of 3.56 rounds lasts as long as one of four (400 000 x 5 kb at band 150 in pieces
    // of 1.53 + 3 x 3.56 rounds: 78 ms of kernels for 12.2 rounds of work; in 1 + 4 + 4 + 3.2: thirteen rounds)
    // (GAMDP_CHUNK_MIN, the tests' way to send small batches through four pieces on two contexts: the plain split)
    const size_t round = (size_t)c->n_cu * 16 * (b150 ? 8 : 2);
    auto whole = [&](size_t v, bool up) { const size_t r = up ? (v + round - 1) / round : v / round; return std::max<size_t>(1, r) * round; };
    const bool by_rounds = chunk_env < 0;
    const size_t pieces = 4, n0 = by_rounds ? std::min(n, whole(n / first_div, false)) : n / first_div;
    const size_t per = by_rounds ? whole((n - n0 + 2) / 3, true) : (n - n0 + 2) / 3;
    size_t bound[5] = {0, n0, std::min(n, n0 + per), std::min(n, n0 + 2 * per), n};
    int rc[2] = {0, 0};
    auto worker = [&](int t) noexcept {
        for (size_t k = (size_t)t; k < pieces && rc[t] == 0; k += 2) {
            const size_t first = bound[k], cnt = bound[k + 1] - bound[k];
            cc[t]->log_piece = (u32)k;
            if (cnt) rc[t] = guarded(cc[t], [&] { return run(cc[t], first, cnt); });
        }
    };
    {
        Threads pool;   // joined on every path out
        pool.start(worker, 1);
        worker(0);
    }
    c->kernel_ms += cc[1]->kernel_ms; c->kernel_launches += cc[1]->kernel_launches;
    c->launch_log.insert(c->launch_log.end(), cc[1]->launch_log.begin(), cc[1]->launch_log.end());
    std::stable_sort(c->launch_log.begin(), c->launch_log.end(), [](const gamdp_launch_info& x, const gamdp_launch_info& y) { return x.piece < y.piece; });
    if (rc[1]) c->set_error(cc[1]->err);
    return rc[0] ? rc[0] : rc[1];
    });
}

int gamdp_task_preflight(uint64_t alen, uint64_t blen, uint32_t band, uint64_t begin_a, uint64_t end_a, uint64_t begin_b,
                         uint64_t end_b, int force_start, int force_end, uint64_t* cells)
{
    u64 X = 0, c = 0;
    const int st = preflight(alen, blen, band, begin_a, end_a, begin_b, end_b, force_start != 0, force_end != 0, &X, &c);
    if (cells) *cells = c;
    return st;
}

unsigned gamdp_build_info(void) { return diag().build ? 1u : 0u; }

void gamdp_encode(const char* chars, uint64_t n, uint8_t* codes)
{
    for (uint64_t i = 0; i < n; i++) codes[i] = encode_char(chars[i]);
}

void gamdp_decode(const uint8_t* codes, uint64_t n, char* chars)
{  // operator char(), nucleotide.code.hpp:111-126
    static const char L[5] = {'A', 'T', 'C', 'G', 'N'};
    for (uint64_t i = 0; i < n; i++) chars[i] = L[codes[i] > 4 ? 4 : codes[i]];
}

void gamdp_revcomp(uint8_t* codes, uint64_t n)
{  // complement then reverse, contig.code.hpp:187-229
    static const uint8_t comp[5] = {1, 0, 3, 2, 4};
    for (uint64_t i = 0; i < n / 2; i++) {
        const uint8_t x = comp[codes[i] > 4 ? 4 : codes[i]], y = comp[codes[n - 1 - i] > 4 ? 4 : codes[n - 1 - i]];
        codes[i] = y;
        codes[n - 1 - i] = x;
    }
    if (n & 1) codes[n / 2] = comp[codes[n / 2] > 4 ? 4 : codes[n / 2]];
}

int64_t gamdp_find_hits(const uint8_t* a, uint64_t alen, uint64_t a_start, uint64_t a_end, const uint8_t* b, uint64_t blen,
                        uint64_t b_start, uint64_t b_end, uint64_t word, uint32_t* hits, uint64_t cap)
{
    std::vector<uint32_t> h;
    find_hits(a, alen, a_start, a_end, b, blen, b_start, b_end, word, h);
    for (size_t i = 0; i < h.size() && i < cap && hits; i++) hits[i] = h[i];
    return (int64_t)h.size();
}

uint64_t gamdp_synth_pair(uint64_t k, uint64_t len, uint8_t* master, uint8_t* slave)
{
    // splitmix64 stream keyed by k; see SURVEY.md section 8(d) for the workload definition
    return synth_pair_codes(k, len, master, slave);
}

int gamdp_seqset_create_synth(gamdp_ctx* ctx, uint64_t first_pair, uint32_t n_pairs, uint64_t len, gamdp_seqset** out)
{
    return gamdp_seqset_create_synth_strided(ctx, first_pair, 1, n_pairs, len, out);
}

int gamdp_seqset_create_synth_strided(gamdp_ctx* ctx, uint64_t first_pair, uint64_t stride_pairs, uint32_t n_pairs, uint64_t len,
                                      gamdp_seqset** out)
{
    if (!ctx || !out || len == 0 || len >= (1ull << 30) || n_pairs >= (1u << 30)) return GAMDP_EINVAL;
    *out = nullptr;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (hipSetDevice(c->device) != hipSuccess) return GAMDP_EHIP;
    SeqSet* s = new (std::nothrow) SeqSet();
    if (!s) return GAMDP_ENOMEM;
    const int rc_ = guarded(c, [&] { return s->upload_synth(c, first_pair, stride_pairs, n_pairs, len); });
    if (rc_) { delete s; return rc_; }
    *out = reinterpret_cast<gamdp_seqset*>(s);
    return 0;
}

}  // extern "C"
