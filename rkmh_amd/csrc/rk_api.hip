// rk_api.hip -- the C ABI of include/rkmh_amd.h on top of the gfx950 kernels (rk_kernels.hip).
// Host-side orchestration only: device memory, tile descriptors, the reference index build, the
// pinned double-buffered H2D/D2H pipeline.  No CPU implementation of any hashing/sketching step lives
// here: every entry point fails with RK_ERR_HIP when no GPU is usable.
#include "../../include/rkmh_amd.h"
#include "rk_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

using namespace rk;

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(RK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define RKCHK(expr) do { int _r = (expr); if (_r != RK_OK) return _r; } while (0)

extern "C" const char* rk_last_error(void) { return g_err.c_str(); }
extern "C" void rk__set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* rk_version(void) { return "rkmh_amd 0.1 (gfx950)"; }
extern "C" void rk_default_policy(rk_policy* p) {
    p->fold = RK_FOLD_SWAP32; p->drop_last_window = 1; p->counter_counts_zero = 1;
    p->mask_strict_less = 1; p->freq_max_inclusive = 1; p->seed = 42;
}
extern "C" int rk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" void rk__pool_forget(void* p); // rk_parse.cpp: big parser buffers are tracked for recycling
extern "C" void rk_free(void* p) { rk__pool_forget(p); free(p); }

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return RK_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(RK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return RK_OK;
    }
    void release() { if (p) { hipError_t e = hipFree(p); (void)e; } p = nullptr; cap = 0; }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return RK_OK;
        if (p) { hipError_t e = hipHostFree(p); (void)e; p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes + 256, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(RK_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes + 256;
        return RK_OK;
    }
    void release() { if (p) { hipError_t e = hipHostFree(p); (void)e; } p = nullptr; cap = 0; }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

struct rk_counter {
    rk_ctx* ctx;
    int32_t* d;
    uint64_t slots;
    uint64_t entries = 0; // int32 entries behind d: `slots` for a full table, the tracked slots of a compact one
    // compact depth map (rk_counter_create_compact): entry e counts the windows whose hash % slots is the e-th smallest of the
    // slots that some key of the context's reference index maps to; key_sid[key id] = its entry
    bool compact = false;
    uint64_t index_gen = 0; // the reference index (rk_ctx::index_gen) the tracked slots were taken from
    DevBuf c_pre, c_tab, c_keysid;
    std::vector<uint32_t> h_tab; // host copy of the (slot, entry) table (rk_counter_get)
    CompactSlots cs{};
    bool owned;
    int device; // copy of ctx->device: destroying a counter after its context must not touch the freed context
    // the slot-partitioned count pass (rk_count.hip) adds to the table with plain read-modify-writes: passes into one table are
    // chained (each waits for `last` on its stream), and they share the scratch arrays
    DevBuf ws;
    hipEvent_t last = nullptr;        // the latest slot-partitioned pass (plain stores): every later pass waits for it
    bool last_set = false;
    hipEvent_t last_atomic = nullptr; // the latest atomic-form pass: only a slot-partitioned pass has to wait for it
    bool last_atomic_set = false;
    std::mutex mu;
};
// every other reader / writer of a table first waits for the count passes enqueued so far
static int counter_settle(const rk_counter* k) {
    if (k && k->last_set) HIPCHK(hipEventSynchronize(k->last));
    if (k && k->last_atomic_set) HIPCHK(hipEventSynchronize(k->last_atomic));
    return RK_OK;
}

struct Slot { // one half of the double-buffered classify pipeline
    PinBuf h_bases, h_offs, h_out;
    DevBuf d_bases, d_offs, d_out;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    int64_t first = 0, n = 0;
    bool busy = false;
};

struct rk_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    DevPolicy pol{};
    // references
    int nref = 0, S = 0;
    KsArr ks{};
    std::vector<uint64_t> h_sk;
    std::vector<int32_t> h_lens;
    DevBuf d_fpb, d_base, d_kv, d_post, d_pre, d_keepbits, d_kpost, d_kbase, d_kkeys, d_kslots;
    DevBuf d_kf4[KM_MAX_KS], d_km1[KM_MAX_KS], d_km1v[KM_MAX_KS]; // k-mer-space structures, one set per k-mer size
    KmerSets ksets{};
    uint32_t kpre_inserted = 0; // k-mers the enumeration found for the k-mer-space structures (diagnostic)
    bool kmer_form_allowed = true; // rk_set_kmer_form
    // rk_set_kmer_cache: the enumeration of the 4^k k-mer universe behind the k-mer-space structures (k_enum_kmers: 26 ms at k = 16,
    // 0.4 s at k = 18) is a function of (the index keys, k, fold, seed) alone -- kept in this file between runs
    std::string kmer_cache_path;
    int kmer_cache_state = 0; // of the last index build: 0 no file given, 1 loaded, 2 enumerated and written, 3 enumerated (the file could not be written)
    RefIndex ix{};
    bool have_refs = false;
    double density = 1.0; // fraction of a reference's k-mers that its sketch keeps (largest over references)
    std::mutex general_mu; // the general path (rerouted rows) works in the context's own buffers: FASTQ slots take turns
    // -M
    rk_counter* depth = nullptr;
    int min_occ = 0;
    // -M with a bounded min_num (rk_set_min_num_bound): < 0 exact (row field 3 = min_num); >= 0: row field 3 = min(min_num, bound),
    // the mask is applied per index KEY (d_keepkey; the k-mer-space kernel reads the masked map copies d_km1m) and no window
    // outside the index is looked up in the depth map except by the probe that counts the first `bound` survivors
    int min_num_bound = -1;
    uint32_t nkeys = 0;                      // distinct sketch hashes = key ids of the index
    std::vector<uint64_t> h_keyhash;         // [nkeys] the hash of each key id (compact depth maps are laid out from it)
    uint64_t index_gen = 0;                  // bumped by every index build: compact depth maps belong to one index
    DevBuf d_keepkey, d_kvm, d_km1m[KM_MAX_KS], d_km1cells[KM_MAX_KS];
    uint32_t km1_ncells[KM_MAX_KS] = {0}, km1_vmask[KM_MAX_KS] = {0};
    KmerSets ksets_m{};                      // ksets with km1 = the masked copies (valid while a bounded depth filter is set)
    // workspaces for the general path
    DevBuf w_bases, w_tiles, w_hashes, w_segoff, w_ids, w_sk, w_lens, w_out, w_misc, w_sel, w_selstate, w_table, w_gcount, w_tail;
    int ref_count_mode = 0; // -I counter fill: 0 per k-mer occurrence (stream), 1 once per distinct hash per reference (filter)
    Slot slot[2];
};

static int set_dev(rk_ctx* c) { HIPCHK(hipSetDevice(c->device)); return RK_OK; }

extern "C" int rk_device_props(int device, int32_t* compute_units, int32_t* clock_khz, int64_t* l2_bytes, int64_t* hbm_bytes) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return fail(RK_ERR_HIP, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (l2_bytes) *l2_bytes = p.l2CacheSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return RK_OK;
}

extern "C" int rk_ctx_create(int device, const rk_policy* policy, rk_ctx** out) {
    if (!out) return fail(RK_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(RK_ERR_HIP, "no HIP device available (%s): rkmh_amd has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RK_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    rk_ctx* c = new rk_ctx();
    c->device = device;
    rk_policy p;
    if (policy) p = *policy; else rk_default_policy(&p);
    c->pol.fold = p.fold; c->pol.drop_last_window = p.drop_last_window;
    c->pol.counter_counts_zero = p.counter_counts_zero; c->pol.mask_strict_less = p.mask_strict_less;
    c->pol.freq_max_inclusive = p.freq_max_inclusive; c->pol.seed = p.seed;
    HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    for (auto& s : c->slot) {
        HIPCHK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    *out = c;
    return RK_OK;
}
extern "C" int rk_ctx_policy(const rk_ctx* c, rk_policy* out) {
    if (!c || !out) return fail(RK_ERR_ARG, "bad arguments");
    out->fold = c->pol.fold; out->drop_last_window = c->pol.drop_last_window; out->counter_counts_zero = c->pol.counter_counts_zero;
    out->mask_strict_less = c->pol.mask_strict_less; out->freq_max_inclusive = c->pol.freq_max_inclusive; out->seed = c->pol.seed;
    return RK_OK;
}
extern "C" void rk_ctx_destroy(rk_ctx* c) {
    if (!c) return;
    hipError_t e = hipSetDevice(c->device); (void)e;
    e = hipDeviceSynchronize(); (void)e;
    for (DevBuf* b : {&c->d_fpb, &c->d_base, &c->d_kv, &c->d_post, &c->d_pre, &c->d_keepbits, &c->d_kpost, &c->d_kbase, &c->d_kkeys, &c->d_kslots, &c->w_bases, &c->w_tiles, &c->w_hashes, &c->w_segoff,
                      &c->w_ids, &c->w_sk, &c->w_lens, &c->w_out, &c->w_misc, &c->w_sel, &c->w_selstate, &c->w_table, &c->w_gcount, &c->w_tail}) b->release();
    for (int j = 0; j < KM_MAX_KS; ++j) { c->d_kf4[j].release(); c->d_km1[j].release(); c->d_km1v[j].release(); c->d_km1m[j].release(); c->d_km1cells[j].release(); }
    c->d_keepkey.release(); c->d_kvm.release();
    for (auto& s : c->slot) {
        s.h_bases.release(); s.h_offs.release(); s.h_out.release();
        s.d_bases.release(); s.d_offs.release(); s.d_out.release();
        if (s.done) { e = hipEventDestroy(s.done); (void)e; }
        if (s.st) { e = hipStreamDestroy(s.st); (void)e; }
    }
    if (c->st) { e = hipStreamDestroy(c->st); (void)e; }
    delete c;
}
extern "C" void* rk_ctx_stream(rk_ctx* c) { return c ? (void*)c->st : nullptr; }
extern "C" int rk_ctx_synchronize(rk_ctx* c) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    RKCHK(set_dev(c));
    HIPCHK(hipDeviceSynchronize());
    return RK_OK;
}

static int check_ks(const int* ks, int nks, KsArr* out) {
    if (!ks || nks < 1 || nks > RK_MAX_KS) return fail(RK_ERR_ARG, "need 1..%d k-mer sizes, got %d", RK_MAX_KS, nks);
    out->n = nks;
    for (int i = 0; i < nks; ++i) {
        if (ks[i] < 1 || ks[i] > RK_MAX_K) return fail(RK_ERR_LIMIT, "k=%d outside [1,%d]", ks[i], RK_MAX_K);
        out->k[i] = ks[i];
    }
    return RK_OK;
}

// ------------------------------------------------------------------------------------------------
// General path: hash tiles -> (optional) in-LDS sort / sketch / intersect, for sequences of any length.
struct GeneralOut {
    uint64_t* hashes = nullptr;      // host, [total hashes of the batch] (caller sized via hash_offsets)
    uint64_t* sketches = nullptr;    // host [n*S]
    int32_t* lens = nullptr;         // host [n]
    int32_t* out4 = nullptr;         // host [n*4]
    bool write_back_sorted = false;  // hashes out = sorted segments (minhashes in-place semantics)
    int32_t* tail_counts = nullptr;  // host [n * (nref - argmax_n)] (cfg.argmax_n > 0)
};
struct GeneralCfg {
    KsArr ks;
    int S = 0;
    rk_counter* inc_counter = nullptr; // increment while hashing (6-arg calc_hashes)
    rk_counter* distinct_counter = nullptr; // increment once per distinct hash per sequence (filter, rkmh.cpp:348-355)
    const DepthTable* depth_insert = nullptr; // call: count every hash of the batch in the exact depth map
    const DepthTable* depth_lookup = nullptr; // call: depth of every hash of the batch -> depth_out[cursor...]
    int32_t* depth_out = nullptr;
    const rk_counter* filt_counter = nullptr;
    int filter_mode = FILTER_NONE, fmin = 0, fmax = 0;
    bool single_kmer = false;          // calc_hash(string): exactly one window of len bases per sequence
    bool classify = false;
    int argmax_n = 0;                  // > 0: argmax over the first argmax_n references only, counts of the rest -> tail_counts
    bool keep_all = false;             // every hash takes part (no bottom-S): sequences with more hashes than S are refused
    // resident batches only (d_bases_in != nullptr): sequence i starts at byte abs_starts[i] of d_bases_in and `offsets`
    // is just the prefix sum of the lengths -- lets a scattered subset of a resident batch run without gathering bases
    const uint64_t* abs_starts = nullptr;
};

static uint32_t next_pow2(uint32_t x) { uint32_t p = 64; while (p < x) p <<= 1; return p; }

// mask_by_frequency of the general path: by slot of the depth table, or -- compact depth map -- through the keep bits of the index keys
static void apply_depth_cfg(const rk_ctx* c, GeneralCfg& cfg) {
    if (!c->depth) return;
    if (c->depth->compact) { cfg.filter_mode = FILTER_KEYMASK; return; }
    cfg.filt_counter = c->depth; cfg.filter_mode = FILTER_MASK_MIN; cfg.fmin = c->min_occ;
}

// d_bases: device pointer to the batch's bases when already resident (else nullptr => upload from `bases`)
// memcpy into a pinned staging buffer with a few threads: one core copies ~14 GB/s, the link takes several times that
static int host_threads() { // workers for the host-side copies and per-read loops (RKMH_COPY_THREADS; default: up to 8 of the CPUs granted)
    static const int nt = []() {
        const char* e = getenv("RKMH_COPY_THREADS");
        int v = e ? atoi(e) : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
        return v < 1 ? 1 : (v > 32 ? 32 : v);
    }();
    return nt;
}
// f(begin, end) over [0, n) in contiguous pieces on host_threads() threads (the caller's thread takes the first piece)
template <typename F>
static void par_for(size_t n, size_t min_piece, F f) {
    const int nt = host_threads();
    if (n < 2 * min_piece || nt == 1) { f((size_t)0, n); return; }
    size_t pieces = std::min<size_t>((size_t)nt, n / min_piece);
    const size_t per = (n + pieces - 1) / pieces;
    std::vector<std::thread> th;
    for (size_t i = 1; i < pieces; ++i) {
        const size_t lo = per * i, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=] { f(lo, hi); });
    }
    f((size_t)0, std::min(n, per));
    for (auto& t : th) t.join();
}
static void par_memcpy(void* dst, const void* src, size_t n) {
    par_for(n, (size_t)4 << 20, [=](size_t lo, size_t hi) { memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
}
// is [p, p + bytes) page-locked host memory the DMA engines can read directly (rk_host_alloc, hipHostMalloc, hipHostRegister)?
static bool is_pinned_host(const void* p, size_t bytes) {
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    hipPointerAttribute_t b;
    if (hipPointerGetAttributes(&b, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
    return b.type == hipMemoryTypeHost;
}

// Page-locks a caller's pageable buffer for the duration of one call (hipHostRegister of memory that is already touched takes
// 2-3 ms per 256 MB on the MI355X host, measured: tools/ubench/host_register.hip -- a tenth of the staging copy it replaces), so the
// DMA engines read / write it in place.  Fails quietly (read-only mappings, a page shared with another registration, a platform
// limit): the caller then takes the staging path.  RKMH_HOST_REGISTER=0 disables.
// Several host threads may work on neighbouring pieces of ONE caller buffer (bin/rkmh --devices: a context per device, each with its
// range of the reads): their page-rounded registrations would overlap, and a piece whose first and last byte lie in a neighbour's
// registered pages would LOOK page-locked (is_pinned_host samples both ends) while its middle is not.  So temporary registrations
// are kept in a process-wide list: a range that touches another thread's registration is neither registered nor trusted.
static std::mutex g_temp_reg_mu;
static std::vector<std::pair<uintptr_t, uintptr_t>> g_temp_reg; // [lo, hi) page ranges registered by a ScopedHostRegister that is alive
static bool touches_temp_registration(const void* p, size_t bytes) { // caller holds g_temp_reg_mu
    const uintptr_t lo = (uintptr_t)p & ~(uintptr_t)4095, hi = ((uintptr_t)p + bytes + 4095) & ~(uintptr_t)4095;
    for (const auto& r : g_temp_reg) if (lo < r.second && r.first < hi) return true;
    return false;
}
// is_pinned_host for a caller's buffer: page-locked by the caller, not merely overlapping a temporary registration of ours
static bool caller_pinned_host(const void* p, size_t bytes) {
    if (!p || bytes == 0) return false;
    std::lock_guard<std::mutex> l(g_temp_reg_mu);
    return !touches_temp_registration(p, bytes) && is_pinned_host(p, bytes);
}
struct ScopedHostRegister {
    uintptr_t lo = 0, hi = 0;
    bool ok = false;
    static bool enabled() {
        static const bool on = [] { const char* e = getenv("RKMH_HOST_REGISTER"); return !(e && *e == '0'); }();
        return on;
    }
    ScopedHostRegister(const void* p, size_t bytes, size_t min_bytes) {
        if (!p || bytes < min_bytes || !enabled()) return;
        std::lock_guard<std::mutex> l(g_temp_reg_mu);
        if (touches_temp_registration(p, bytes)) return; // a neighbouring piece of the same buffer is registered: staging path
        lo = (uintptr_t)p & ~(uintptr_t)4095; hi = ((uintptr_t)p + bytes + 4095) & ~(uintptr_t)4095;
        if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterPortable) == hipSuccess) { ok = true; g_temp_reg.emplace_back(lo, hi); }
        else (void)hipGetLastError();
    }
    ~ScopedHostRegister() {
        if (!ok) return;
        std::lock_guard<std::mutex> l(g_temp_reg_mu);
        hipError_t e = hipHostUnregister((void*)lo); (void)e;
        for (size_t i = 0; i < g_temp_reg.size(); ++i) if (g_temp_reg[i].first == lo && g_temp_reg[i].second == hi) { g_temp_reg.erase(g_temp_reg.begin() + (long)i); break; }
    }
    ScopedHostRegister(const ScopedHostRegister&) = delete;
    ScopedHostRegister& operator=(const ScopedHostRegister&) = delete;
};

// Host -> device copy of a pageable buffer through the context's two pinned staging buffers (the same ones the fused
// host pipeline uses): the CPU fills one while the DMA engine drains the other.  hipMemcpyAsync straight from pageable
// memory runs at a fraction of the link rate and blocks the caller for the whole transfer.
static int upload_staged(rk_ctx* c, void* dst, const uint8_t* src, size_t bytes, hipStream_t st) {
    const size_t CH = 16u << 20;
    // (a few megabytes -- a reference panel -- are not worth two 16 MB page-locked buffers: creating those takes longer than the copy)
    if (bytes <= (4u << 20)) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); return RK_OK; }
    for (int i = 0; i < 2; ++i) {
        if (c->slot[i].busy) { HIPCHK(hipEventSynchronize(c->slot[i].done)); c->slot[i].busy = false; }
        RKCHK(c->slot[i].h_bases.reserve(CH));
    }
    int which = 0;
    bool used[2] = {false, false};
    for (size_t off = 0; off < bytes; off += CH) {
        const size_t nb = bytes - off < CH ? bytes - off : CH;
        Slot& sl = c->slot[which];
        if (used[which]) HIPCHK(hipEventSynchronize(sl.done)); // its previous chunk has left the pinned buffer
        par_memcpy(sl.h_bases.p, src + off, nb);
        HIPCHK(hipMemcpyAsync((uint8_t*)dst + off, sl.h_bases.p, nb, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(sl.done, st));
        used[which] = true;
        which ^= 1;
    }
    return RK_OK;
}

static int general_run(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases_in, const uint64_t* offsets, int64_t n,
                       const GeneralCfg& cfg, const GeneralOut& out) {
    RKCHK(set_dev(c));
    if (n <= 0) return RK_OK;
    const bool need_sort = out.sketches || out.lens || out.out4 || out.write_back_sorted;
    if ((cfg.inc_counter && cfg.inc_counter->compact) || (cfg.distinct_counter && cfg.distinct_counter->compact) || (cfg.filt_counter && cfg.filt_counter->compact))
        return fail(RK_ERR_STATE, "a compact depth map only serves rk_count_batch* of reads that fit the sketch and rk_set_depth_filter");
    if (cfg.classify && !c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    const uint64_t MAX_CHUNK_BASES = 1ull << 28, MAX_CHUNK_HASHES = 1ull << 26;
    std::vector<TileDesc> tiles;
    std::vector<uint64_t> seg;
    std::vector<std::vector<uint32_t>> classes(32);
    std::vector<uint32_t> long_seqs, presel;
    const uint64_t PRESEL_MAX_HASHES = 1ull << 18; // one block streams its sequence a few times; beyond this the multi-block select is faster (measured: 3 M hashes 4 ms vs 0.7 ms)
    uint64_t hash_cursor = 0; // position in out.hashes
    int64_t i0 = 0;
    static const bool gtiming = getenv("RKMH_INDEX_TIMING") != nullptr; // (stderr: where a general-path batch spends its time)
    auto gt0 = std::chrono::steady_clock::now();
    auto gtick = [&](const char* what) {
        if (!gtiming) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rkmh general] %-24s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - gt0).count());
        gt0 = now;
    };
    while (i0 < n) {
        // ---- pick a chunk [i0,i1)
        int64_t i1 = i0;
        uint64_t cb = 0, ch = 0;
        tiles.clear(); seg.clear(); seg.push_back(0);
        for (auto& v : classes) v.clear();
        long_seqs.clear();
        const uint64_t base0 = offsets[i0];
        while (i1 < n) {
            uint64_t len = offsets[i1 + 1] - offsets[i1];
            uint64_t nh = 0;
            if (cfg.single_kmer) nh = 1;
            else for (int j = 0; j < cfg.ks.n; ++j) nh += (uint64_t)num_windows((int)len, cfg.ks.k[j], c->pol.drop_last_window);
            if (len > 0x7fffffffull) return fail(RK_ERR_LIMIT, "sequence %lld longer than 2^31-1", (long long)i1);
            if (cfg.filter_mode == FILTER_KEYMASK && (nh > (uint64_t)cfg.S || cfg.keep_all))
                return fail(RK_ERR_NEED_FULL, "sequence %lld has %llu hashes for a sketch of %d: bottom-s selection needs the depth of every hash, "
                            "which a compact depth map does not hold", (long long)i1, (unsigned long long)nh, cfg.S);
            if (cfg.keep_all && nh > (uint64_t)cfg.S)
                return fail(RK_ERR_LIMIT, "sequence %lld has %llu hashes; without bottom-s selection at most %d take part", (long long)i1,
                            (unsigned long long)nh, cfg.S);
            if (i1 > i0 && (cb + len > MAX_CHUNK_BASES || ch + nh > MAX_CHUNK_HASHES)) break;
            if (need_sort && nh > (uint64_t)SORT_MAX_P && out.write_back_sorted)
                return fail(RK_ERR_LIMIT, "sequence %lld has %llu hashes; in-place sorting handles <= %d",
                            (long long)i1, (unsigned long long)nh, SORT_MAX_P);
            // tiles
            uint64_t o = seg.back();
            uint64_t rel = cfg.abs_starts ? cfg.abs_starts[i1] : offsets[i1] - base0;
            if (cfg.single_kmer) {
                if (len < 1 || len > RK_MAX_K) return fail(RK_ERR_LIMIT, "k-mer length %llu outside [1,%d]", (unsigned long long)len, RK_MAX_K);
                tiles.push_back(TileDesc{rel, o, (uint32_t)len, 1u, (uint32_t)len, 0u});
                o += 1;
            } else {
                for (int j = 0; j < cfg.ks.n; ++j) {
                    int k = cfg.ks.k[j];
                    uint32_t nw = (uint32_t)num_windows((int)len, k, c->pol.drop_last_window);
                    for (uint32_t w0 = 0; w0 < nw; w0 += HASH_TILE_WIN) {
                        uint32_t cnt = std::min<uint32_t>(HASH_TILE_WIN, nw - w0);
                        tiles.push_back(TileDesc{rel + w0, o + w0, cnt + (uint32_t)k - 1u, cnt, (uint32_t)k, 0u});
                    }
                    o += nw;
                }
            }
            seg.push_back(o);
            if (need_sort) {
                if (nh > (uint64_t)SORT_MAX_P) long_seqs.push_back((uint32_t)(i1 - i0)); // radix select, then sort <= S candidates
                else {
                    uint32_t P = next_pow2((uint32_t)nh);
                    int cls = 0; while ((64u << cls) < P) ++cls;
                    classes[cls].push_back((uint32_t)(i1 - i0));
                }
            }
            cb += len; ch += nh; ++i1;
        }
        const int64_t cn = i1 - i0;
        gtick("chunk planned");
        // ---- upload
        const uint8_t* d_bases;
        if (d_bases_in) d_bases = cfg.abs_starts ? d_bases_in : d_bases_in + base0;
        else {
            RKCHK(c->w_bases.reserve(cb + 64));
            if (cb) RKCHK(upload_staged(c, c->w_bases.p, bases + base0, cb, c->st));
            d_bases = c->w_bases.as<uint8_t>();
        }
        if (((uintptr_t)d_bases & 3) != 0) {
            // stage_piece reads aligned dwords; a misaligned base pointer is folded into the tile offsets
            uint64_t mis = (uintptr_t)d_bases & 3;
            d_bases -= mis;
            for (auto& t : tiles) t.base_off += mis;
        }
        RKCHK(c->w_tiles.reserve(tiles.size() * sizeof(TileDesc)));
        RKCHK(c->w_segoff.reserve(seg.size() * 8));
        RKCHK(c->w_hashes.reserve((ch + 1) * 8));
        if (!tiles.empty()) HIPCHK(hipMemcpyAsync(c->w_tiles.p, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, c->st));
        HIPCHK(hipMemcpyAsync(c->w_segoff.p, seg.data(), seg.size() * 8, hipMemcpyHostToDevice, c->st));
        HIPCHK(launch_hash_tiles(d_bases, c->w_tiles.as<TileDesc>(), (uint32_t)tiles.size(), c->w_hashes.as<uint64_t>(),
                                 cfg.inc_counter ? cfg.inc_counter->d : nullptr, cfg.inc_counter ? cfg.inc_counter->slots : 1,
                                 c->pol, c->st));
        gtick("uploaded, hashing launched");
        if (out.hashes && !out.write_back_sorted && ch)
            HIPCHK(hipMemcpyAsync(out.hashes + hash_cursor, c->w_hashes.p, ch * 8, hipMemcpyDeviceToHost, c->st));
        if (cfg.depth_insert) HIPCHK(launch_depth_insert(c->w_hashes.as<uint64_t>(), ch, *cfg.depth_insert, c->st));
        if (cfg.depth_lookup) HIPCHK(launch_depth_lookup(c->w_hashes.as<uint64_t>(), ch, *cfg.depth_lookup, cfg.depth_out + hash_cursor, c->st));
        if (cfg.distinct_counter) {
            for (int64_t q = 0; q < cn; ++q) {
                const uint64_t n_h = seg[(size_t)q + 1] - seg[(size_t)q];
                if (n_h == 0) continue;
                uint64_t tsize = 1024;
                while (tsize < 2 * n_h) tsize <<= 1;
                RKCHK(c->w_table.reserve((tsize + 1) * 8));
                HIPCHK(launch_count_distinct(c->w_hashes.as<uint64_t>() + seg[(size_t)q], n_h, c->w_table.as<uint64_t>(), tsize,
                                             cfg.distinct_counter->d, cfg.distinct_counter->slots, c->st));
            }
        }
        if (need_sort) {
            const int S = cfg.S;
            if (out.sketches) RKCHK(c->w_sk.reserve((size_t)cn * S * 8));
            if (out.lens) RKCHK(c->w_lens.reserve((size_t)cn * 4));
            if (out.out4) RKCHK(c->w_out.reserve((size_t)cn * 16));
            const size_t ntail = (cfg.classify && cfg.argmax_n > 0 && out.tail_counts) ? (size_t)(c->ix.nref - cfg.argmax_n) : 0;
            if (ntail) RKCHK(c->w_tail.reserve((size_t)cn * ntail * 4));
            RKCHK(c->w_ids.reserve((size_t)cn * 4));
            size_t id_cursor = 0;
            // panels whose per-reference counter row does not fit the LDS beside the largest sort buffer count in global rows
            int32_t* gcount = nullptr;
            uint32_t gcount_rows = 0;
            if (cfg.classify && out.out4) {
                // with classification every launch sorts at most next_pow2(S) values (longer sequences are pre-selected)
                gcount_rows = sort_intersect_global_rows(std::max<uint32_t>(64u, next_pow2((uint32_t)S)), c->ix.nref);
                if (gcount_rows) {
                    RKCHK(c->w_gcount.reserve((size_t)gcount_rows * (size_t)c->ix.nref * 4));
                    gcount = c->w_gcount.as<int32_t>();
                }
            }
            auto sort_args = [&](uint32_t* d_ids, uint32_t count, uint32_t P) {
                SortArgs a{};
                a.hashes = c->w_hashes.as<uint64_t>(); a.seg_off = c->w_segoff.as<uint64_t>();
                a.seq_ids = d_ids; a.nlist = count; a.P = P; a.S = S;
                a.write_back = out.write_back_sorted ? 1 : 0;
                a.sketches = out.sketches ? c->w_sk.as<uint64_t>() : nullptr;
                a.lens = out.lens ? c->w_lens.as<int32_t>() : nullptr;
                a.out4 = out.out4 ? c->w_out.as<int32_t>() : nullptr;
                a.counter = cfg.filt_counter ? cfg.filt_counter->d : nullptr;
                a.slots = cfg.filt_counter ? cfg.filt_counter->slots : 1;
                a.filter_mode = cfg.filter_mode; a.fmin = cfg.fmin; a.fmax = cfg.fmax;
                if (cfg.classify && out.out4) { a.gcount = gcount; a.gcount_rows = gcount_rows; }
                if (cfg.classify) { a.argmax_n = cfg.argmax_n; a.tail_counts = ntail ? c->w_tail.as<int32_t>() : nullptr; }
                return a;
            };
            // Sequences with far more hashes than the sketch keeps (long reads, genomes up to a few million k-mers) are not
            // sorted whole: their block radix-selects the bottom S first and sorts only those.
            const uint32_t Psel = std::max<uint32_t>(64u, next_pow2((uint32_t)S));
            const bool can_presel = !out.write_back_sorted && Psel <= (uint32_t)SORT_MAX_P;
            presel.clear();
            for (int cls = 0; cls < 32; ++cls) {
                auto& ids = classes[cls];
                if (ids.empty()) continue;
                if (can_presel && (64u << cls) > Psel) { presel.insert(presel.end(), ids.begin(), ids.end()); continue; }
                uint32_t* d_ids = c->w_ids.as<uint32_t>() + id_cursor;
                HIPCHK(hipMemcpyAsync(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->st));
                id_cursor += ids.size();
                SortArgs a = sort_args(d_ids, (uint32_t)ids.size(), 64u << cls);
                HIPCHK(launch_sort_intersect(a, cfg.classify ? &c->ix : nullptr, c->pol, c->st));
            }
            if (can_presel) { // long sequences of moderate size take the same route; only the huge ones need the multi-block select
                size_t keep = 0;
                for (uint32_t li : long_seqs) {
                    if (seg[li + 1] - seg[li] <= PRESEL_MAX_HASHES) presel.push_back(li);
                    else long_seqs[keep++] = li;
                }
                long_seqs.resize(keep);
            }
            if (!presel.empty()) {
                uint32_t* d_ids = c->w_ids.as<uint32_t>() + id_cursor;
                HIPCHK(hipMemcpyAsync(d_ids, presel.data(), presel.size() * 4, hipMemcpyHostToDevice, c->st));
                id_cursor += presel.size();
                SortArgs a = sort_args(d_ids, (uint32_t)presel.size(), Psel);
                a.preselect = 1;
                HIPCHK(launch_sort_intersect(a, cfg.classify ? &c->ix : nullptr, c->pol, c->st));
            }
            if (!long_seqs.empty()) { // sequences longer than the LDS sorter: exact bottom-S by radix select first
                RKCHK(c->w_sel.reserve((size_t)S * 8 + 64));
                RKCHK(c->w_selstate.reserve(16 * 4 + 8192 * 4));
                uint32_t* st_ = c->w_selstate.as<uint32_t>();
                for (uint32_t li : long_seqs) {
                    uint32_t* d_id = c->w_ids.as<uint32_t>() + id_cursor;
                    HIPCHK(hipMemcpyAsync(d_id, &li, 4, hipMemcpyHostToDevice, c->st));
                    id_cursor += 1;
                    const uint64_t n_h = seg[li + 1] - seg[li];
                    HIPCHK(launch_select_bottom(c->w_hashes.as<uint64_t>() + seg[li], n_h, S,
                                                cfg.filt_counter ? cfg.filt_counter->d : nullptr, cfg.filt_counter ? cfg.filt_counter->slots : 1,
                                                cfg.filter_mode, cfg.fmin, cfg.fmax, c->pol, st_, st_ + 16, c->w_sel.as<uint64_t>(), c->st));
                    SortArgs a{};
                    a.hashes = c->w_hashes.as<uint64_t>(); a.seg_off = c->w_segoff.as<uint64_t>();
                    a.seq_ids = d_id; a.nlist = 1; a.P = next_pow2((uint32_t)S); a.S = S; a.write_back = 0;
                    a.sketches = out.sketches ? c->w_sk.as<uint64_t>() : nullptr;
                    a.lens = out.lens ? c->w_lens.as<int32_t>() : nullptr;
                    a.out4 = out.out4 ? c->w_out.as<int32_t>() : nullptr;
                    a.filter_mode = FILTER_NONE;
                    if (cfg.classify && out.out4) { a.gcount = gcount; a.gcount_rows = gcount_rows; }
                    if (cfg.classify) { a.argmax_n = cfg.argmax_n; a.tail_counts = ntail ? c->w_tail.as<int32_t>() : nullptr; }
                    a.sel_hashes = c->w_sel.as<uint64_t>(); a.sel_len = st_ + 8;
                    HIPCHK(launch_sort_intersect(a, cfg.classify ? &c->ix : nullptr, c->pol, c->st));
                    HIPCHK(hipStreamSynchronize(c->st)); // w_sel / state are reused by the next long sequence
                }
            }
            gtick("sorts launched");
            // the ids vectors must outlive the async copies
            HIPCHK(hipStreamSynchronize(c->st));
            gtick("kernels done");
            if (out.write_back_sorted && out.hashes && ch)
                HIPCHK(hipMemcpyAsync(out.hashes + hash_cursor, c->w_hashes.p, ch * 8, hipMemcpyDeviceToHost, c->st));
            if (out.sketches) HIPCHK(hipMemcpyAsync(out.sketches + (size_t)i0 * S, c->w_sk.p, (size_t)cn * S * 8, hipMemcpyDeviceToHost, c->st));
            if (out.lens) HIPCHK(hipMemcpyAsync(out.lens + i0, c->w_lens.p, (size_t)cn * 4, hipMemcpyDeviceToHost, c->st));
            if (out.out4) HIPCHK(hipMemcpyAsync(out.out4 + (size_t)i0 * 4, c->w_out.p, (size_t)cn * 16, hipMemcpyDeviceToHost, c->st));
            if (ntail) HIPCHK(hipMemcpyAsync(out.tail_counts + (size_t)i0 * ntail, c->w_tail.p, (size_t)cn * ntail * 4, hipMemcpyDeviceToHost, c->st));
        }
        HIPCHK(hipStreamSynchronize(c->st));
        gtick("results downloaded");
        // -M with a bounded min_num: the general path computes min_num exactly; rows carry min(min_num, bound) on every path
        if (out.out4 && cfg.classify && !cfg.keep_all && (cfg.filter_mode == FILTER_MASK_MIN || cfg.filter_mode == FILTER_KEYMASK) && c->min_num_bound >= 0)
            for (int64_t q = i0; q < i1; ++q) if (out.out4[q * 4 + 3] > c->min_num_bound) out.out4[q * 4 + 3] = c->min_num_bound;
        hash_cursor += ch;
        i0 = i1;
    }
    return RK_OK;
}

static void fill_hash_offsets(const rk_ctx* c, const uint64_t* offsets, int64_t n, const KsArr& ks, uint64_t* ho) {
    ho[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t len = offsets[i + 1] - offsets[i], nh = 0;
        for (int j = 0; j < ks.n; ++j) nh += (uint64_t)num_windows((int)len, ks.k[j], c->pol.drop_last_window);
        ho[i + 1] = ho[i] + nh;
    }
}

// ------------------------------------------------------------------------------------------------
// inner boundary
extern "C" int rk_to_upper(rk_ctx* c, char* seq, int len) {
    if (!c || (!seq && len > 0) || len < 0) return fail(RK_ERR_ARG, "bad arguments");
    if (len == 0) return RK_OK;
    RKCHK(set_dev(c));
    RKCHK(c->w_bases.reserve((size_t)len));
    HIPCHK(hipMemcpyAsync(c->w_bases.p, seq, (size_t)len, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_to_upper(c->w_bases.as<uint8_t>(), (uint64_t)len, c->st));
    HIPCHK(hipMemcpyAsync(seq, c->w_bases.p, (size_t)len, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return RK_OK;
}

static int calc_hashes_impl(rk_ctx* c, const char* seq, int len, const int* ks, int nks, uint64_t** out, int* n, rk_counter* counter) {
    if (!c || !out || !n || (!seq && len > 0) || len < 0) return fail(RK_ERR_ARG, "bad arguments");
    GeneralCfg cfg;
    RKCHK(check_ks(ks, nks, &cfg.ks));
    cfg.inc_counter = counter;
    uint64_t offs[2] = {0, (uint64_t)len};
    uint64_t ho[2];
    fill_hash_offsets(c, offs, 1, cfg.ks, ho);
    uint64_t* h = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(ho[1] ? ho[1] : 1));
    if (!h) return fail(RK_ERR_NOMEM, "malloc");
    // upper-casing is the caller's job in the reference (to_upper precedes calc_hashes, rkmh.cpp:856-860);
    // the device upper-cases on the fly, which is idempotent for already upper-cased input.
    GeneralOut go; go.hashes = h;
    int r = general_run(c, (const uint8_t*)seq, nullptr, offs, 1, cfg, go);
    if (r != RK_OK) { free(h); return r; }
    *out = h; *n = (int)ho[1];
    return RK_OK;
}
extern "C" int rk_calc_hashes(rk_ctx* c, const char* seq, int len, const int* ks, int nks, uint64_t** out, int* n) {
    return calc_hashes_impl(c, seq, len, ks, nks, out, n, nullptr);
}
extern "C" int rk_calc_hashes_counted(rk_ctx* c, const char* seq, int len, const int* ks, int nks, uint64_t** out, int* n, rk_counter* counter) {
    if (!counter) return fail(RK_ERR_ARG, "counter is NULL");
    return calc_hashes_impl(c, seq, len, ks, nks, out, n, counter);
}
extern "C" int rk_calc_hash(rk_ctx* c, const char* kmer, int k, uint64_t* out) {
    if (!c || !kmer || !out) return fail(RK_ERR_ARG, "bad arguments");
    GeneralCfg cfg; cfg.single_kmer = true; cfg.ks.n = 1; cfg.ks.k[0] = k;
    uint64_t offs[2] = {0, (uint64_t)k};
    GeneralOut go; go.hashes = out;
    return general_run(c, (const uint8_t*)kmer, nullptr, offs, 1, cfg, go);
}

// sort-only pipeline over hashes that are already on the host (minhashes & friends)
static int minhashes_impl(rk_ctx* c, uint64_t* h, int n, int S, uint64_t** mins, int* m, const rk_counter* counter,
                          int filter_mode, int fmin, int fmax, bool sort_input) {
    if (!c || (!h && n > 0) || n < 0 || !mins || !m) return fail(RK_ERR_ARG, "bad arguments");
    if (S < 1 || S > RK_MAX_SKETCH) return fail(RK_ERR_LIMIT, "sketch size %d outside [1,%d]", S, RK_MAX_SKETCH);
    RKCHK(set_dev(c));
    uint64_t* r = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)S);
    if (!r) return fail(RK_ERR_NOMEM, "malloc");
    int rc = RK_OK;
    if (n > SORT_MAX_P) do {
        // Longer than the in-LDS sorter holds (the reference calls minhashes on every whole reference, rkmh.cpp:822, :835-836):
        // the sketch is the exact bottom S of the kept hashes by radix select (the route rk_set_references takes for long
        // sequences) + a sort of those <= S values; the side effect of mkmh::minhashes -- the caller's array comes back sorted
        // ascending -- is a whole-array device sort (rk_sort.hip).
        size_t tmp_bytes = 0;
        hipError_t e = sort_input ? sort_u64_temp_bytes((uint64_t)n, &tmp_bytes) : hipSuccess;
        if (e != hipSuccess) { rc = fail(RK_ERR_HIP, "minhashes (long input): %s", hipGetErrorString(e)); break; }
        if ((rc = c->w_hashes.reserve((size_t)(n + 1) * 8)) != RK_OK) break;
        if ((rc = c->w_segoff.reserve(16)) != RK_OK) break;
        if ((rc = c->w_ids.reserve(4)) != RK_OK) break;
        if ((rc = c->w_sk.reserve((size_t)S * 8)) != RK_OK) break;
        if ((rc = c->w_lens.reserve(4)) != RK_OK) break;
        if ((rc = c->w_sel.reserve((size_t)S * 8 + 64)) != RK_OK) break;
        if ((rc = c->w_selstate.reserve(16 * 4 + 8192 * 4)) != RK_OK) break;
        if (sort_input && (rc = c->w_misc.reserve(tmp_bytes)) != RK_OK) break;
        uint64_t seg[2] = {0, (uint64_t)n};
        uint32_t id0 = 0;
        int32_t len = 0;
        uint32_t* st_ = c->w_selstate.as<uint32_t>();
        e = hipMemcpyAsync(c->w_hashes.p, h, (size_t)n * 8, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(c->w_segoff.p, seg, 16, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(c->w_ids.p, &id0, 4, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = launch_select_bottom(c->w_hashes.as<uint64_t>(), (uint64_t)n, S, counter ? counter->d : nullptr, counter ? counter->slots : 1,
                                                      filter_mode, fmin, fmax, c->pol, st_, st_ + 16, c->w_sel.as<uint64_t>(), c->st);
        SortArgs a{};
        a.hashes = c->w_hashes.as<uint64_t>(); a.seg_off = c->w_segoff.as<uint64_t>(); a.seq_ids = c->w_ids.as<uint32_t>();
        a.nlist = 1; a.P = std::max<uint32_t>(64u, next_pow2((uint32_t)S)); a.S = S; a.write_back = 0;
        a.sketches = c->w_sk.as<uint64_t>(); a.lens = c->w_lens.as<int32_t>(); a.out4 = nullptr;
        a.filter_mode = FILTER_NONE; // the selection already applied the filter
        a.sel_hashes = c->w_sel.as<uint64_t>(); a.sel_len = st_ + 8;
        if (e == hipSuccess) e = launch_sort_intersect(a, nullptr, c->pol, c->st);
        if (e == hipSuccess && sort_input) e = launch_sort_u64(c->w_hashes.as<uint64_t>(), (uint64_t)n, c->w_misc.p, tmp_bytes, c->st);
        if (e == hipSuccess && sort_input) e = hipMemcpyAsync(h, c->w_hashes.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(r, c->w_sk.p, (size_t)S * 8, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(&len, c->w_lens.p, 4, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipStreamSynchronize(c->st);
        if (e != hipSuccess) { rc = fail(RK_ERR_HIP, "minhashes pipeline (long input): %s", hipGetErrorString(e)); break; }
        *m = len;
    } while (0);
    else do {
        if ((rc = c->w_hashes.reserve((size_t)(n + 1) * 8)) != RK_OK) break;
        if ((rc = c->w_segoff.reserve(16)) != RK_OK) break;
        if ((rc = c->w_ids.reserve(4)) != RK_OK) break;
        if ((rc = c->w_sk.reserve((size_t)S * 8)) != RK_OK) break;
        if ((rc = c->w_lens.reserve(4)) != RK_OK) break;
        uint64_t seg[2] = {0, (uint64_t)n};
        uint32_t id0 = 0;
        int32_t len = 0;
        hipError_t e = hipSuccess;
        if (n) e = hipMemcpyAsync(c->w_hashes.p, h, (size_t)n * 8, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(c->w_segoff.p, seg, 16, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(c->w_ids.p, &id0, 4, hipMemcpyHostToDevice, c->st);
        SortArgs a{};
        a.hashes = c->w_hashes.as<uint64_t>(); a.seg_off = c->w_segoff.as<uint64_t>(); a.seq_ids = c->w_ids.as<uint32_t>();
        a.nlist = 1; a.P = next_pow2((uint32_t)n); a.S = S; a.write_back = sort_input ? 1 : 0;
        a.sketches = c->w_sk.as<uint64_t>(); a.lens = c->w_lens.as<int32_t>(); a.out4 = nullptr;
        a.counter = counter ? counter->d : nullptr; a.slots = counter ? counter->slots : 1;
        a.filter_mode = filter_mode; a.fmin = fmin; a.fmax = fmax;
        if (e == hipSuccess) e = launch_sort_intersect(a, nullptr, c->pol, c->st);
        if (e == hipSuccess && sort_input && n) e = hipMemcpyAsync(h, c->w_hashes.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(r, c->w_sk.p, (size_t)S * 8, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(&len, c->w_lens.p, 4, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipStreamSynchronize(c->st);
        if (e != hipSuccess) { rc = fail(RK_ERR_HIP, "minhashes pipeline: %s", hipGetErrorString(e)); break; }
        *m = len;
    } while (0);
    if (rc != RK_OK) { free(r); return rc; }
    *mins = r;
    return RK_OK;
}
extern "C" int rk_minhashes(rk_ctx* c, uint64_t* h, int n, int S, uint64_t** mins, int* m) {
    return minhashes_impl(c, h, n, S, mins, m, nullptr, FILTER_NONE, 0, 0, true);
}
extern "C" int rk_minhashes_frequency_filter(rk_ctx* c, uint64_t* h, int n, int S, uint64_t** out, int* m,
                                             const rk_counter* counter, int min_count, int max_count) {
    if (!counter) return fail(RK_ERR_ARG, "counter is NULL");
    return minhashes_impl(c, h, n, S, out, m, counter, FILTER_RANGE, min_count, max_count, true);
}
extern "C" int rk_mask_by_frequency(rk_ctx* c, uint64_t* h, int n, const rk_counter* counter, int min_occ) {
    if (!c || !counter || (!h && n > 0) || n < 0) return fail(RK_ERR_ARG, "bad arguments");
    if (n == 0) return RK_OK;
    RKCHK(set_dev(c));
    RKCHK(c->w_hashes.reserve((size_t)n * 8));
    HIPCHK(hipMemcpyAsync(c->w_hashes.p, h, (size_t)n * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_mask_by_frequency(c->w_hashes.as<uint64_t>(), (uint64_t)n, counter->d, counter->slots, min_occ, c->pol, c->st));
    HIPCHK(hipMemcpyAsync(h, c->w_hashes.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return RK_OK;
}

extern "C" int rk_hash_intersection_size(rk_ctx* c, const uint64_t* a, int na, const uint64_t* b, int nb, int* out) {
    if (!c || !out || na < 0 || nb < 0 || (!a && na) || (!b && nb)) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(c));
    RKCHK(c->w_hashes.reserve((size_t)(na + nb + 2) * 8));
    RKCHK(c->w_lens.reserve(4));
    uint64_t* da = c->w_hashes.as<uint64_t>();
    uint64_t* db = da + na;
    if (na) HIPCHK(hipMemcpyAsync(da, a, (size_t)na * 8, hipMemcpyHostToDevice, c->st));
    if (nb) HIPCHK(hipMemcpyAsync(db, b, (size_t)nb * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_intersect_pair(da, na, db, nb, c->w_lens.as<int>(), c->st));
    HIPCHK(hipMemcpyAsync(out, c->w_lens.p, 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    return RK_OK;
}

// mkmh::hash_intersection with 7 arguments, as filter's classify_* helpers call it (equiv.hpp:308, 340, 364): (array, start,
// length) twice, then the sketch size, which bounds the result.  Callee-allocated result, released with rk_free.
extern "C" int rk_hash_intersection(rk_ctx* c, const uint64_t* a, int a_start, int a_len, const uint64_t* b, int b_start, int b_len,
                                    int S, uint64_t** out, int* n) {
    if (!c || !out || !n || a_start < 0 || b_start < 0 || a_len < 0 || b_len < 0 || S < 0 || (!a && a_len) || (!b && b_len))
        return fail(RK_ERR_ARG, "bad arguments");
    *out = nullptr; *n = 0;
    RKCHK(set_dev(c));
    const int cap = S < a_len ? S : a_len;
    RKCHK(c->w_hashes.reserve((size_t)(a_len + b_len + cap + 2) * 8));
    RKCHK(c->w_lens.reserve(4));
    uint64_t* da = c->w_hashes.as<uint64_t>();
    uint64_t* db = da + a_len;
    uint64_t* dout = db + b_len;
    if (a_len) HIPCHK(hipMemcpyAsync(da, a + a_start, (size_t)a_len * 8, hipMemcpyHostToDevice, c->st));
    if (b_len) HIPCHK(hipMemcpyAsync(db, b + b_start, (size_t)b_len * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_intersect_pair_emit(da, a_len, db, b_len, cap, dout, c->w_lens.as<int>(), c->st));
    int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, c->w_lens.p, 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    uint64_t* r = (uint64_t*)malloc((size_t)(cnt > 0 ? cnt : 1) * 8);
    if (!r) return fail(RK_ERR_NOMEM, "malloc");
    if (cnt) HIPCHK(hipMemcpy(r, dout, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    *out = r; *n = cnt;
    return RK_OK;
}

// ---- HASHTCounter ------------------------------------------------------------------------------
extern "C" int rk_counter_create(rk_ctx* c, uint64_t slots, rk_counter** out) {
    if (!c || !out || slots == 0) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(c));
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, slots * 4);
    if (e != hipSuccess) return fail(RK_ERR_NOMEM, "hipMalloc(%llu) for counter: %s", (unsigned long long)(slots * 4), hipGetErrorString(e));
    HIPCHK(hipMemsetAsync(d, 0, slots * 4, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    rk_counter* k = new rk_counter();
    k->ctx = c; k->d = (int32_t*)d; k->slots = slots; k->entries = slots; k->owned = true; k->device = c->device;
    *out = k;
    return RK_OK;
}
extern "C" int rk_counter_wrap(rk_ctx* c, void* d, uint64_t slots, rk_counter** out) {
    if (!c || !out || !d || slots == 0) return fail(RK_ERR_ARG, "bad arguments");
    rk_counter* k = new rk_counter();
    k->ctx = c; k->d = (int32_t*)d; k->slots = slots; k->entries = slots; k->owned = false; k->device = c->device;
    *out = k;
    return RK_OK;
}
extern "C" void rk_counter_destroy(rk_counter* k) {
    if (!k) return;
    hipError_t e = hipSetDevice(k->device); (void)e;
    if (k->last_set) { e = hipEventSynchronize(k->last); (void)e; }
    if (k->last_atomic_set) { e = hipEventSynchronize(k->last_atomic); (void)e; }
    if (k->last) { e = hipEventDestroy(k->last); (void)e; }
    if (k->last_atomic) { e = hipEventDestroy(k->last_atomic); (void)e; }
    k->ws.release(); k->c_pre.release(); k->c_tab.release(); k->c_keysid.release();
    if (k->owned) { e = hipFree(k->d); (void)e; }
    delete k;
}
extern "C" int rk_counter_clear(rk_counter* k) {
    if (!k) return fail(RK_ERR_ARG, "counter is NULL");
    RKCHK(set_dev(k->ctx));
    RKCHK(counter_settle(k));
    HIPCHK(hipMemsetAsync(k->d, 0, k->entries * 4, k->ctx->st));
    HIPCHK(hipStreamSynchronize(k->ctx->st));
    return RK_OK;
}
// dst += src (element-wise) and dst = src for two tables of the same size that may live on different devices / contexts: the
// reduce and broadcast steps of a multi-device -M run inside one process (one rk_ctx per device; the reference's OpenMP threads
// share ONE HASHTCounter instead, src/rkmh.cpp:739,909).  A table on another device is brought over in 64 MB pieces.
static int counter_combine(rk_counter* dst, const rk_counter* src, bool add) {
    if (!dst || !src) return fail(RK_ERR_ARG, "counter is NULL");
    if (dst->slots != src->slots) return fail(RK_ERR_ARG, "counters of %llu and %llu slots", (unsigned long long)dst->slots, (unsigned long long)src->slots);
    if (dst->compact != src->compact || dst->entries != src->entries)
        return fail(RK_ERR_ARG, "a compact and a full depth map, or compact maps of different reference sets, cannot be combined");
    if (dst == src || dst->d == src->d) return add ? fail(RK_ERR_ARG, "rk_counter_add of a table to itself") : RK_OK;
    RKCHK(set_dev(src->ctx));
    RKCHK(counter_settle(src));
    HIPCHK(hipStreamSynchronize(src->ctx->st)); // whatever filled src on its own context's stream is complete
    RKCHK(set_dev(dst->ctx));
    RKCHK(counter_settle(dst));
    hipStream_t st = dst->ctx->st;
    if (!add) { HIPCHK(hipMemcpyAsync(dst->d, src->d, src->entries * 4, hipMemcpyDefault, st)); HIPCHK(hipStreamSynchronize(st)); return RK_OK; }
    // RKMH_COUNTER_STAGED=1 takes the staged branch below even for two tables of ONE device (a one-GPU box can test it)
    static const bool force_staged = getenv("RKMH_COUNTER_STAGED") && atoi(getenv("RKMH_COUNTER_STAGED")) != 0;
    if (dst->device == src->device && !force_staged) { HIPCHK(launch_counter_add(dst->d, src->d, dst->entries, st)); HIPCHK(hipStreamSynchronize(st)); return RK_OK; }
    if (dst->device != src->device && !force_staged) {
        // two devices of one node: with peer access the add kernel reads the other device's table in place over xGMI
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
            hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);
            if (pe == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); pe = hipSuccess; }
            if (pe == hipSuccess) { HIPCHK(launch_counter_add(dst->d, src->d, dst->entries, st)); HIPCHK(hipStreamSynchronize(st)); return RK_OK; }
            (void)hipGetLastError();
        }
    }
    // staged: the other table comes over in 64 MB pieces (hipMemcpyDefault device -> device) on a copy stream, two buffers, so that
    // piece i + 1 is in flight while piece i is being added
    const uint64_t CH = (uint64_t)16 << 20; // slots per piece
    DevBuf tmp[2];
    int rc = RK_OK;
    for (int i = 0; i < 2 && rc == RK_OK; ++i) rc = tmp[i].reserve(std::min<uint64_t>(CH, dst->entries) * 4);
    hipEvent_t copied[2] = {nullptr, nullptr}, added[2] = {nullptr, nullptr};
    hipStream_t cst = nullptr;
    if (rc == RK_OK && hipStreamCreateWithFlags(&cst, hipStreamNonBlocking) != hipSuccess) rc = fail(RK_ERR_HIP, "hipStreamCreate failed");
    for (int i = 0; i < 2 && rc == RK_OK; ++i)
        if (hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&added[i], hipEventDisableTiming) != hipSuccess)
            rc = fail(RK_ERR_HIP, "hipEventCreate failed");
    int which = 0;
    uint64_t piece = 0;
    for (uint64_t off = 0; off < dst->entries && rc == RK_OK; off += CH, which ^= 1, ++piece) {
        const uint64_t n = std::min<uint64_t>(CH, dst->entries - off);
        hipError_t e = hipSuccess;
        if (piece >= 2) e = hipStreamWaitEvent(cst, added[which], 0);           // the buffer's previous piece has been added
        if (e == hipSuccess) e = hipMemcpyAsync(tmp[which].p, src->d + off, n * 4, hipMemcpyDefault, cst);
        if (e == hipSuccess) e = hipEventRecord(copied[which], cst);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, copied[which], 0);
        if (e == hipSuccess) e = launch_counter_add(dst->d + off, tmp[which].as<int32_t>(), n, st);
        if (e == hipSuccess) e = hipEventRecord(added[which], st);
        if (e != hipSuccess) rc = fail(RK_ERR_HIP, "rk_counter_add: %s", hipGetErrorString(e));
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == RK_OK) rc = fail(RK_ERR_HIP, "rk_counter_add: synchronize failed");
    if (cst) { hipError_t e = hipStreamSynchronize(cst); (void)e; e = hipStreamDestroy(cst); (void)e; }
    for (int i = 0; i < 2; ++i) {
        if (copied[i]) { hipError_t e = hipEventDestroy(copied[i]); (void)e; }
        if (added[i]) { hipError_t e = hipEventDestroy(added[i]); (void)e; }
        tmp[i].release();
    }
    return rc;
}
extern "C" int rk_counter_add(rk_counter* dst, const rk_counter* src) { return counter_combine(dst, src, true); }
extern "C" int rk_counter_copy(rk_counter* dst, const rk_counter* src) { return counter_combine(dst, src, false); }
static int not_for_compact(const rk_counter* k, const char* what) {
    return (k && k->compact) ? fail(RK_ERR_STATE, "%s: a compact depth map only counts whole batches (rk_count_batch*) and only the slots of index keys", what) : RK_OK;
}
extern "C" int rk_counter_increment(rk_counter* k, uint64_t key) {
    if (!k) return fail(RK_ERR_ARG, "counter is NULL");
    RKCHK(not_for_compact(k, "rk_counter_increment"));
    RKCHK(set_dev(k->ctx));
    RKCHK(counter_settle(k));
    HIPCHK(launch_counter_inc(k->d, k->slots, key, k->ctx->st));
    HIPCHK(hipStreamSynchronize(k->ctx->st));
    return RK_OK;
}
extern "C" int rk_counter_get(const rk_counter* k, uint64_t key, int32_t* out) {
    if (!k || !out) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(k->ctx));
    RKCHK(counter_settle(k));
    if (k->compact) { // the entry of the key's slot, if that slot is tracked (it is for every index key)
        const uint32_t s32 = (uint32_t)(key % k->slots);
        for (uint32_t idx = (s32 * 0x85EBCA6Bu) >> k->cs.tab_shift;; idx = (idx + 1u) & k->cs.tab_mask) {
            if (k->h_tab[2 * (size_t)idx] == s32) { HIPCHK(hipMemcpy(out, k->d + k->h_tab[2 * (size_t)idx + 1], 4, hipMemcpyDeviceToHost)); return RK_OK; }
            if (k->h_tab[2 * (size_t)idx] == CS_EMPTY) return fail(RK_ERR_STATE, "rk_counter_get: the key's slot is not tracked by this compact depth map");
        }
    }
    HIPCHK(hipMemcpy(out, k->d + (key % k->slots), 4, hipMemcpyDeviceToHost));
    return RK_OK;
}
// Depth-map files.  "RKHT2\n", u64 slots, u64 nnz, u32 tag_len, tag bytes, then nnz x (u32 slot, i32 count).  The tag is an
// opaque provenance record (rk_depth_map_tag: k list, hash policy, fingerprint of the read set); a file saved with a tag
// only loads when the caller presents the identical tag, so a map counted from other reads or under another hashing policy is
// refused instead of silently producing wrong masks.  "RKHT1\n" files (round 1: no tag field) still load as untagged.
static int counter_save_impl(rk_counter* k, const char* path, const void* tag, uint32_t tag_len) {
    if (!k || !path || (tag_len && !tag)) return fail(RK_ERR_ARG, "bad arguments");
    if (tag_len > 4096) return fail(RK_ERR_ARG, "tag too long");
    if (k->slots > 0xffffffffull) return fail(RK_ERR_LIMIT, "counter too large to serialise (slot index is 32 bit)");
    RKCHK(not_for_compact(k, "rk_counter_save"));
    RKCHK(set_dev(k->ctx));
    RKCHK(counter_settle(k));
    std::vector<int32_t> h((size_t)k->slots);
    HIPCHK(hipMemcpy(h.data(), k->d, k->slots * 4, hipMemcpyDeviceToHost));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(RK_ERR_IO, "cannot write %s", path);
    uint64_t nnz = 0;
    for (int32_t v : h) nnz += v != 0;
    bool ok = fwrite("RKHT2\n", 1, 6, f) == 6 && fwrite(&k->slots, 8, 1, f) == 1 && fwrite(&nnz, 8, 1, f) == 1 &&
              fwrite(&tag_len, 4, 1, f) == 1 && (tag_len == 0 || fwrite(tag, 1, tag_len, f) == tag_len);
    std::vector<uint32_t> rec;
    rec.reserve(1 << 16);
    for (size_t i = 0; ok && i < h.size(); ++i) {
        if (h[i] == 0) continue;
        rec.push_back((uint32_t)i); rec.push_back((uint32_t)h[i]);
        if (rec.size() >= (1 << 16)) { ok = fwrite(rec.data(), 4, rec.size(), f) == rec.size(); rec.clear(); }
    }
    if (ok && !rec.empty()) ok = fwrite(rec.data(), 4, rec.size(), f) == rec.size();
    ok = (fclose(f) == 0) && ok;
    return ok ? RK_OK : fail(RK_ERR_IO, "short write to %s", path);
}
static int counter_load_impl(rk_counter* k, const char* path, const void* tag, uint32_t tag_len) {
    if (!k || !path || (tag_len && !tag)) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(not_for_compact(k, "rk_counter_load"));
    RKCHK(set_dev(k->ctx));
    RKCHK(counter_settle(k));
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RK_ERR_IO, "cannot read %s", path);
    char magic[6];
    uint64_t slots = 0, nnz = 0;
    uint32_t flen = 0;
    bool ok = fread(magic, 1, 6, f) == 6;
    const bool v1 = ok && memcmp(magic, "RKHT1\n", 6) == 0, v2 = ok && memcmp(magic, "RKHT2\n", 6) == 0;
    ok = (v1 || v2) && fread(&slots, 8, 1, f) == 1 && fread(&nnz, 8, 1, f) == 1 && (v1 || fread(&flen, 4, 1, f) == 1) && flen <= 4096;
    std::vector<uint8_t> ftag(flen);
    if (ok && flen) ok = fread(ftag.data(), 1, flen, f) == flen;
    if (!ok) { fclose(f); return fail(RK_ERR_IO, "%s is not a counter file", path); }
    if (flen != tag_len || (flen && memcmp(ftag.data(), tag, flen) != 0)) {
        fclose(f);
        if (flen == 0) return fail(RK_ERR_ARG, "%s carries no provenance tag: refusing to use it as the depth map of these reads", path);
        if (tag_len == 0) return fail(RK_ERR_ARG, "%s carries a provenance tag: load it with rk_counter_load_tagged", path);
        return fail(RK_ERR_ARG, "%s was counted from other reads, k-mer sizes or hashing policy than this run (provenance tag mismatch): refusing to load it", path);
    }
    if (slots != k->slots) { fclose(f); return fail(RK_ERR_ARG, "%s holds %llu slots, the counter has %llu", path, (unsigned long long)slots, (unsigned long long)k->slots); }
    std::vector<int32_t> h((size_t)slots, 0);
    std::vector<uint32_t> rec(1 << 16);
    uint64_t left = nnz * 2;
    while (ok && left) {
        size_t want = left < rec.size() ? (size_t)left : rec.size();
        ok = fread(rec.data(), 4, want, f) == want;
        for (size_t i = 0; ok && i + 1 < want; i += 2) { if (rec[i] >= slots) { ok = false; break; } h[rec[i]] = (int32_t)rec[i + 1]; }
        left -= want;
    }
    fclose(f);
    if (!ok) return fail(RK_ERR_IO, "%s is truncated or corrupt", path);
    HIPCHK(hipMemcpy(k->d, h.data(), slots * 4, hipMemcpyHostToDevice));
    return RK_OK;
}
extern "C" int rk_counter_save(rk_counter* k, const char* path) { return counter_save_impl(k, path, nullptr, 0); }
extern "C" int rk_counter_load(rk_counter* k, const char* path) { return counter_load_impl(k, path, nullptr, 0); }
extern "C" int rk_counter_save_tagged(rk_counter* k, const char* path, const void* tag, uint32_t tag_len) {
    return counter_save_impl(k, path, tag, tag_len);
}
extern "C" int rk_counter_load_tagged(rk_counter* k, const char* path, const void* tag, uint32_t tag_len) {
    return counter_load_impl(k, path, tag, tag_len);
}
// Provenance of a read-depth map: everything that decides which slot a read's k-mers increment (k list, seed, fold, window and
// zero-counting policy) plus a fingerprint of the read set (count, total bases, FNV-1a over the read lengths and over up to
// 2 x 1 MiB of bases from both ends of the batch).
extern "C" int rk_depth_map_tag(const rk_ctx* c, const int* ks, int nks, const uint8_t* bases, const uint64_t* offsets,
                                int64_t nseq, uint8_t tag[RK_DEPTH_TAG_BYTES]) {
    if (!c || !ks || nks < 1 || nks > RK_MAX_KS || !offsets || nseq < 0 || !tag || (nseq > 0 && !bases)) return fail(RK_ERR_ARG, "bad arguments");
    auto fnv = [](uint64_t h, const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
        return h;
    };
    struct Tag { char magic[8]; int32_t fold, drop_last, counts_zero; uint32_t seed; int32_t nks; int32_t ks[RK_MAX_KS]; int64_t nseq; uint64_t total, hlen, hbases; } t;
    static_assert(sizeof(Tag) <= RK_DEPTH_TAG_BYTES, "tag layout");
    memset(&t, 0, sizeof t);
    memcpy(t.magic, "rkdepth2", 8); // 2: the fingerprint covers every base (1 sampled both ends)
    t.fold = c->pol.fold; t.drop_last = c->pol.drop_last_window; t.counts_zero = c->pol.counter_counts_zero; t.seed = c->pol.seed;
    t.nks = nks;
    for (int i = 0; i < nks; ++i) t.ks[i] = ks[i];
    t.nseq = nseq;
    t.total = offsets[nseq] - offsets[0];
    uint64_t h = 0xcbf29ce484222325ull;
    for (int64_t i = 0; i < nseq; ++i) { const uint64_t len = offsets[i + 1] - offsets[i]; h = fnv(h, &len, 8); }
    t.hlen = h;
    // EVERY base takes part (a read set edited in the middle, same lengths, must not look like the one the map was counted from):
    // 64-bit multiply-rotate hash over 8-byte words, 4 MB pieces hashed in parallel and combined in order
    {
        const uint8_t* b0 = bases + offsets[0];
        const size_t total = (size_t)t.total, PIECE = (size_t)4 << 20, npieces = (total + PIECE - 1) / PIECE;
        std::vector<uint64_t> ph(npieces, 0);
        par_for(npieces, 1, [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; ++p) {
                const uint8_t* q = b0 + p * PIECE;
                const size_t n = std::min(PIECE, total - p * PIECE);
                uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
                size_t i = 0;
                for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, q + i, 8); x = (x ^ w) * 0xff51afd7ed558ccdull; x = (x << 29) | (x >> 35); }
                uint64_t w = 0;
                if (i < n) { memcpy(&w, q + i, n - i); x = (x ^ w) * 0xff51afd7ed558ccdull; x = (x << 29) | (x >> 35); }
                ph[p] = x;
            }
        });
        h = 0xcbf29ce484222325ull;
        for (uint64_t x : ph) h = fnv(h, &x, 8);
        t.hbases = h;
    }
    memset(tag, 0, RK_DEPTH_TAG_BYTES);
    memcpy(tag, &t, sizeof t);
    return RK_OK;
}
extern "C" void* rk_counter_device_ptr(rk_counter* k) { return k ? k->d : nullptr; }
extern "C" uint64_t rk_counter_slots(const rk_counter* k) { return k ? k->slots : 0; }
extern "C" uint64_t rk_counter_entries(const rk_counter* k) { return k ? k->entries : 0; }
extern "C" int rk_counter_is_compact(const rk_counter* k) { return k && k->compact ? 1 : 0; }

// The compact depth map of a -M run that only needs min_num up to bound 0 (rk_set_min_num_bound): mask_by_frequency then acts
// through the index keys alone, and whether a key survives depends on ONE slot of the table -- key % slots.  So only those slots
// are counted (pass 1 hashes every window as before, but a window whose slot is not one of them is dropped after one bit test):
// the table shrinks from `slots` int32 (800 MB for the reference's 2 * 10^8, rkmh.cpp:739) to one int32 per distinct tracked slot,
// there is no slot array to bin, and the sum over devices or ranks moves a few hundred KB.
// the tracked slots of the compact map for the reference index of `c` and a table of `slots`, ascending; key_sid[key id] = the
// entry that counts the key's slot.  Deterministic in (index, slots): every rank and device of a run lays its map out identically.
static int compact_layout(const rk_ctx* c, uint64_t slots, std::vector<uint32_t>& islots, std::vector<uint32_t>* key_sid) {
    if (!c->have_refs) return fail(RK_ERR_STATE, "a compact depth map is laid out from the reference index: call rk_set_references first");
    if (slots == 0 || slots > 0xFFFFFFFFull) return fail(RK_ERR_LIMIT, "compact depth map: slots must be at most 2^32 - 1");
    islots.resize(c->nkeys);
    for (uint32_t j = 0; j < c->nkeys; ++j) islots[j] = (uint32_t)(c->h_keyhash[j] % slots);
    std::vector<uint32_t> sorted(islots);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    if (key_sid) {
        key_sid->resize(c->nkeys);
        for (uint32_t j = 0; j < c->nkeys; ++j) (*key_sid)[j] = (uint32_t)(std::lower_bound(sorted.begin(), sorted.end(), islots[j]) - sorted.begin());
    }
    islots.swap(sorted);
    return RK_OK;
}
extern "C" int rk_counter_compact_entries(const rk_ctx* c, uint64_t slots, uint64_t* entries) {
    if (!c || !entries) return fail(RK_ERR_ARG, "bad arguments");
    std::vector<uint32_t> islots;
    RKCHK(compact_layout(c, slots, islots, nullptr));
    *entries = islots.empty() ? 1 : islots.size();
    return RK_OK;
}
extern "C" int rk_counter_create_compact(rk_ctx* c, uint64_t slots, void* d_counts_int32, rk_counter** out) {
    if (!c || !out) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(c));
    std::vector<uint32_t> islots, key_sid;
    RKCHK(compact_layout(c, slots, islots, &key_sid));
    const size_t m = islots.size(), entries = m ? m : 1;
    // the slot filter: one bit per hashed slot, 32-64 bits per tracked slot, at most 2^23 bits = 1 MB (L2-resident beside the hashing)
    uint32_t pre_lg = 12;
    while (pre_lg < 23 && ((size_t)1 << pre_lg) < m * 64) ++pre_lg;
    std::vector<uint32_t> pre(((size_t)1 << pre_lg) / 32, 0u);
    uint32_t tab_lg = 4;
    while (((size_t)1 << tab_lg) < 2 * m + 2) ++tab_lg;
    std::vector<uint32_t> tab(((size_t)2 << tab_lg), CS_EMPTY);
    const uint32_t pre_shift = 32u - pre_lg, tab_shift = 32u - tab_lg, tab_mask = (1u << tab_lg) - 1u;
    for (size_t e = 0; e < m; ++e) {
        const uint32_t s32 = islots[e], bit = (s32 * 0x9E3779B1u) >> pre_shift;
        pre[bit >> 5] |= 1u << (bit & 31u);
        uint32_t idx = (s32 * 0x85EBCA6Bu) >> tab_shift;
        while (tab[2 * (size_t)idx] != CS_EMPTY) idx = (idx + 1u) & tab_mask;
        tab[2 * (size_t)idx] = s32; tab[2 * (size_t)idx + 1] = (uint32_t)e;
    }
    rk_counter* k = new rk_counter();
    k->ctx = c; k->slots = slots; k->entries = entries; k->compact = true; k->index_gen = c->index_gen; k->device = c->device;
    k->owned = d_counts_int32 == nullptr; k->d = (int32_t*)d_counts_int32;
    int rc = RK_OK;
    if (k->owned) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, entries * 4);
        if (e != hipSuccess) rc = fail(RK_ERR_NOMEM, "hipMalloc(%zu) for the compact depth map: %s", entries * 4, hipGetErrorString(e));
        else { k->d = (int32_t*)d; if (hipMemsetAsync(d, 0, entries * 4, c->st) != hipSuccess) rc = fail(RK_ERR_HIP, "hipMemsetAsync failed"); }
    }
    if (rc == RK_OK) rc = k->c_pre.reserve(pre.size() * 4);
    if (rc == RK_OK) rc = k->c_tab.reserve(tab.size() * 4);
    if (rc == RK_OK) rc = k->c_keysid.reserve(key_sid.size() * 4 + 16);
    if (rc == RK_OK && (hipMemcpyAsync(k->c_pre.p, pre.data(), pre.size() * 4, hipMemcpyHostToDevice, c->st) != hipSuccess ||
                        hipMemcpyAsync(k->c_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->st) != hipSuccess ||
                        (!key_sid.empty() && hipMemcpyAsync(k->c_keysid.p, key_sid.data(), key_sid.size() * 4, hipMemcpyHostToDevice, c->st) != hipSuccess) ||
                        hipStreamSynchronize(c->st) != hipSuccess))
        rc = fail(RK_ERR_HIP, "compact depth map: upload failed");
    if (rc != RK_OK) { rk_counter_destroy(k); return rc; }
    k->cs.pre = k->c_pre.as<uint32_t>(); k->cs.tab = k->c_tab.as<uint2>();
    k->cs.pre_shift = pre_shift; k->cs.tab_shift = tab_shift; k->cs.tab_mask = tab_mask;
    k->h_tab.swap(tab);
    *out = k;
    return RK_OK;
}

// ---- batched: hash / sketch --------------------------------------------------------------------
extern "C" int rk_hash_batch(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nseq,
                             const int* ks, int nks, uint64_t** out, uint64_t* hash_offsets) {
    if (!c || !offsets || nseq < 0 || !out || !hash_offsets) return fail(RK_ERR_ARG, "bad arguments");
    GeneralCfg cfg;
    RKCHK(check_ks(ks, nks, &cfg.ks));
    fill_hash_offsets(c, offsets, nseq, cfg.ks, hash_offsets);
    uint64_t total = hash_offsets[nseq];
    uint64_t* h = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(total ? total : 1));
    if (!h) return fail(RK_ERR_NOMEM, "malloc");
    GeneralOut go; go.hashes = h;
    int r = general_run(c, bases, nullptr, offsets, nseq, cfg, go);
    if (r != RK_OK) { free(h); return r; }
    *out = h;
    return RK_OK;
}

extern "C" int rk_sketch_batch(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nseq,
                               const int* ks, int nks, int S, uint64_t* sketches, int32_t* lens) {
    if (!c || !offsets || nseq < 0 || !sketches || !lens) return fail(RK_ERR_ARG, "bad arguments");
    if (S < 1 || S > RK_MAX_SKETCH) return fail(RK_ERR_LIMIT, "sketch size %d outside [1,%d]", S, RK_MAX_SKETCH);
    GeneralCfg cfg;
    RKCHK(check_ks(ks, nks, &cfg.ks));
    cfg.S = S;
    GeneralOut go; go.sketches = sketches; go.lens = lens;
    return general_run(c, bases, nullptr, offsets, nseq, cfg, go);
}

// ---- references --------------------------------------------------------------------------------
// The posting lists of the k-mer-space kernel (RefIndex::kpost).  `post` holds one list per key; the genomes of one family share
// most of their sketch hashes, so many keys carry the same list and most of the others carry a list that differs from it in a
// few places (BASELINE config 3's panel: 61 near-identical Zika genomes, 21 HPV16 variants -- a read of theirs walked ~640
// postings).  Identical lists are stored once, and up to KBASE_MAX frequent long lists become BASES: a list close to a base is
// stored as (base, exceptions) -- the kernel adds one to the read's counter of that base, applies the few exceptions (+1 for a
// reference the base lacks, -1 for one it has in excess) and expands each touched base once per read before the arg-max
// (k_classify_kmer, phase 2).  Every such list is ALSO kept in plain form (the sparse-counter kernels cannot subtract).
// remap[offset in post] = (offset of the form the dense-counter kernels walk, offset of the plain form), both into kpost.
// kpost list = header (entries | (base + 1) << 24; base field 0: plain) then entries x (reference, multiplicity; bit 31: -1).
// A list within eight exceptions of its base (most of them) needs no list at all: base and exceptions go INTO the compound value
// (ix, iy, iw: tag 111, base, count, eight 10-bit fields of reference and sign) and the lane that finds the hit applies them.
constexpr int KBASE_MAX = 8;
struct KList { uint32_t enc = 0, plain = 0, ix = 0, iy = 0, iw = 0; };
static void build_kpost(const std::vector<uint32_t>& post, int R, std::vector<uint32_t>& kpost, std::vector<uint32_t>& kbase,
                        std::unordered_map<uint32_t, KList>& remap) {
    struct Dist { std::vector<std::pair<uint32_t, uint32_t>> e; uint32_t weight = 0, plain = 0, enc = 0, ix = 0, iy = 0, iw = 0; bool simple = true; };
    std::map<std::vector<std::pair<uint32_t, uint32_t>>, uint32_t> ids; // list content (sorted by reference) -> distinct id
    std::vector<Dist> dl;
    std::vector<std::pair<uint32_t, uint32_t>> owner; // (offset in post, distinct id)
    for (size_t off = 1; off < post.size();) {
        const uint32_t n = post[off];
        std::vector<std::pair<uint32_t, uint32_t>> e(n);
        for (uint32_t q = 0; q < n; ++q) e[q] = {post[off + 1 + 2 * q], post[off + 2 + 2 * q]};
        std::sort(e.begin(), e.end());
        auto it = ids.find(e);
        if (it == ids.end()) {
            it = ids.emplace(e, (uint32_t)dl.size()).first;
            Dist d; d.e = e;
            for (auto& x : e) d.simple = d.simple && x.second == 1u;
            dl.push_back(std::move(d));
        }
        dl[it->second].weight += 1;
        owner.emplace_back((uint32_t)off, it->second);
        off += 1 + 2 * (size_t)n;
    }
    static const int nbase_env = getenv("RKMH_KBASES") ? atoi(getenv("RKMH_KBASES")) : KBASE_MAX;
    const int nbase_max = R <= 0xFFFF ? std::min(std::max(nbase_env, 0), KBASE_MAX) : 0;
    // bases: the heaviest long lists (keys x references) that are not close to a base already chosen
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < dl.size(); ++i) if (dl[i].simple && dl[i].e.size() >= 8) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint64_t wa = (uint64_t)dl[a].weight * dl[a].e.size(), wb = (uint64_t)dl[b].weight * dl[b].e.size();
        return wa != wb ? wa > wb : a < b;
    });
    auto sym_diff = [](const std::vector<std::pair<uint32_t, uint32_t>>& a, const std::vector<uint32_t>& b) {
        size_t i = 0, j = 0, d = 0;
        while (i < a.size() && j < b.size()) { if (a[i].first == b[j]) { ++i; ++j; } else if (a[i].first < b[j]) { ++i; ++d; } else { ++j; ++d; } }
        return d + (a.size() - i) + (b.size() - j);
    };
    std::vector<std::vector<uint32_t>> bases;
    for (uint32_t i : order) {
        if ((int)bases.size() >= nbase_max) break;
        bool far = true;
        for (auto& b : bases) far = far && sym_diff(dl[i].e, b) > std::max<size_t>(4, dl[i].e.size() / 4);
        if (!far) continue;
        std::vector<uint32_t> b;
        for (auto& x : dl[i].e) b.push_back(x.first);
        bases.push_back(std::move(b));
    }
    kbase.assign(2 * KBASE_MAX, 0u);
    for (size_t b = 0; b < bases.size(); ++b) {
        kbase[2 * b] = (uint32_t)kbase.size(); kbase[2 * b + 1] = (uint32_t)bases[b].size();
        kbase.insert(kbase.end(), bases[b].begin(), bases[b].end());
    }
    kbase.resize(kbase.size() + 64, 0u); // (the expansion reads 16 members at a time)
    kpost.assign(1, 0u);
    for (auto& d : dl) {
        d.plain = (uint32_t)kpost.size();
        kpost.push_back((uint32_t)d.e.size());
        for (auto& x : d.e) { kpost.push_back(x.first); kpost.push_back(x.second); }
        d.enc = d.plain;
        if (!d.simple || d.e.size() < 8 || bases.empty()) continue;
        size_t best = 0, bd = ~(size_t)0;
        for (size_t b = 0; b < bases.size(); ++b) { const size_t dd = sym_diff(d.e, bases[b]); if (dd < bd) { bd = dd; best = b; } }
        if (2 * (1 + bd) > d.e.size()) continue; // not worth it: at least half of the walk must go
        d.enc = (uint32_t)kpost.size();
        kpost.push_back((uint32_t)bd | ((uint32_t)(best + 1) << 24));
        const std::vector<uint32_t>& B = bases[best];
        size_t i = 0, j = 0;
        uint32_t ex[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nex = 0;
        bool small_refs = true;
        auto exception = [&](uint32_t ref, bool neg) {
            kpost.push_back(ref); kpost.push_back(neg ? 0x80000001u : 1u);
            small_refs = small_refs && ref < 512u;
            if (nex < 8) ex[nex] = ref | (neg ? 512u : 0u);
            ++nex;
        };
        while (i < d.e.size() || j < B.size()) {
            if (j == B.size() || (i < d.e.size() && d.e[i].first < B[j])) { exception(d.e[i].first, false); ++i; }
            else if (i == d.e.size() || B[j] < d.e[i].first) { exception(B[j], true); ++j; }
            else { ++i; ++j; }
        }
        static const bool inline_ok = !(getenv("RKMH_KBASE_INLINE") && atoi(getenv("RKMH_KBASE_INLINE")) == 0);
        if (inline_ok && nex <= 8 && small_refs) {
            d.ix = 0xE0000000u | ((uint32_t)best << 26) | (nex << 22) | ex[0] | (ex[1] << 10);
            d.iy = ex[2] | (ex[3] << 10) | (ex[4] << 20);
            d.iw = ex[5] | (ex[6] << 10) | (ex[7] << 20);
        }
    }
    kpost.resize(kpost.size() + 64, 0u); // a hit's first sixteen postings are requested with the header
    for (auto& o : owner) { KList kl; kl.enc = dl[o.second].enc; kl.plain = dl[o.second].plain; kl.ix = dl[o.second].ix; kl.iy = dl[o.second].iy; kl.iw = dl[o.second].iw; remap[o.first] = kl; }
}

// ---- the k-mer enumeration cache (rk_set_kmer_cache) ----
// File: "RKKM1\n", u64 tag, u32 entries, then per entry {u32 k, u32 found, found x (u32 k-mer, u32 key id)}.  The tag is a hash of
// everything the lists depend on: every index key in key-id order, the number of keys, fold and seed.  Any other file is ignored
// (and overwritten after the enumeration has run): a cache never changes results, it only skips the work that would reproduce it.
static uint64_t kmer_cache_tag(const rk_ctx* c, const std::vector<uint32_t>& dense, size_t nkeys) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 0x100000001b3ull; h ^= h >> 29; };
    mix(0x726b6b6d31ull); mix((uint64_t)nkeys); mix((uint64_t)(uint32_t)c->pol.fold); mix((uint64_t)c->pol.seed);
    for (size_t q = 0; q < nkeys; ++q) mix(((uint64_t)dense[q * 4 + 1] << 32) | dense[q * 4]);
    return h;
}
static bool kmer_cache_read(const std::string& path, uint64_t tag, std::map<int, std::vector<uint32_t>>& lists) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[6];
    uint64_t ftag = 0;
    uint32_t n = 0;
    bool ok = fread(magic, 1, 6, f) == 6 && memcmp(magic, "RKKM1\n", 6) == 0 && fread(&ftag, 8, 1, f) == 1 && fread(&n, 4, 1, f) == 1 && ftag == tag && n <= 64;
    for (uint32_t i = 0; ok && i < n; ++i) {
        uint32_t k = 0, found = 0;
        ok = fread(&k, 4, 1, f) == 1 && fread(&found, 4, 1, f) == 1 && k >= 1 && k <= (uint32_t)KW_MAX_K && found <= 0x3fffffffu;
        if (!ok) break;
        std::vector<uint32_t> l((size_t)found * (k > 16 ? 3 : 2)); // (k-mer, key id) -- a wide k-mer takes two words
        ok = l.empty() || fread(l.data(), 4, l.size(), f) == l.size();
        if (ok) lists[(int)k] = std::move(l);
    }
    fclose(f);
    if (!ok) lists.clear();
    return ok;
}
static bool kmer_cache_write(const std::string& path, uint64_t tag, const std::map<int, std::vector<uint32_t>>& lists) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const uint32_t n = (uint32_t)lists.size();
    bool ok = fwrite("RKKM1\n", 1, 6, f) == 6 && fwrite(&tag, 8, 1, f) == 1 && fwrite(&n, 4, 1, f) == 1;
    for (auto& kv : lists) {
        const uint32_t k = (uint32_t)kv.first;
        const uint32_t found = (uint32_t)(kv.second.size() / (k > 16 ? 3 : 2));
        ok = ok && fwrite(&k, 4, 1, f) == 1 && fwrite(&found, 4, 1, f) == 1 && (kv.second.empty() || fwrite(kv.second.data(), 4, kv.second.size(), f) == kv.second.size());
    }
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0; // (atomic: a concurrent reader sees the old file or the new one)
    if (!ok) remove(tmp.c_str());
    return ok;
}

static int build_key_mask(rk_ctx* c);
// RKMH_INDEX_TIMING=1: where the time of an index build goes (stderr)
struct IndexClock {
    bool on = getenv("RKMH_INDEX_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void tick(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[rkmh index] %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
static int build_index(rk_ctx* c) {
    IndexClock clk;
    struct Pair { uint64_t h; uint32_t ref; };
    const int R = c->nref, S = c->S;
    if (R > 0xFFFFF) return fail(RK_ERR_LIMIT, "more than 2^20-1 references");
    std::vector<Pair> pairs;
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < c->h_lens[(size_t)r]; ++j) {
            uint64_t h = c->h_sk[(size_t)r * S + j];
            if (h != 0) pairs.push_back(Pair{h, (uint32_t)r});
        }
    // by (hash, reference): the pairs come in reference order, so a stable radix sort on the hash alone (four 16-bit digits) does it
    {
        std::vector<Pair> tmp(pairs.size());
        std::vector<uint32_t> cnt((size_t)1 << 16);
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 16 * pass;
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (const Pair& p : pairs) ++cnt[(size_t)((p.h >> sh) & 0xFFFFu)];
            uint32_t run = 0;
            for (uint32_t& v : cnt) { const uint32_t here = v; v = run; run += here; }
            for (const Pair& p : pairs) tmp[cnt[(size_t)((p.h >> sh) & 0xFFFFu)]++] = p;
            pairs.swap(tmp);
        }
    }
    clk.tick("pairs sorted");
    size_t distinct = 0;
    for (size_t i = 0; i < pairs.size(); ++i) if (i == 0 || pairs[i].h != pairs[i - 1].h) ++distinct;
    // bucketed table: 8 slots per bucket, at most 2.5 keys per bucket on average (P(more than 8) ~ 0.1 %)
    uint32_t nb = 256, lg = 8;
    size_t load_pct = 250;
    if (const char* e = getenv("RKMH_INDEX_LOAD")) { long v = atol(e); if (v >= 10 && v <= 700) load_pct = (size_t)v; }
    while ((size_t)nb * load_pct < distinct * 100 + 100) { nb <<= 1; ++lg; }
    const uint32_t size = nb * IDX_SLOTS;
    std::vector<uint16_t> fpb(size, 0);
    std::vector<uint32_t> kv((size_t)size * 4, 0); // {key lo, key hi, value, 0} per slot
    std::vector<uint32_t> post;
    post.push_back(0);
    const uint32_t bmask = nb - 1, bshift = 32 - lg;
    size_t i = 0;
    std::vector<std::pair<uint32_t, uint32_t>> grp;
    while (i < pairs.size()) {
        size_t j = i;
        grp.clear();
        while (j < pairs.size() && pairs[j].h == pairs[i].h) {
            size_t k = j; while (k < pairs.size() && pairs[k].h == pairs[i].h && pairs[k].ref == pairs[j].ref) ++k;
            grp.emplace_back(pairs[j].ref, (uint32_t)(k - j));
            j = k;
        }
        uint32_t v;
        if (grp.size() == 1 && grp[0].second <= 0x1FFu) v = grp[0].first | (grp[0].second << 20);
        else if (grp.size() == 2 && grp[0].second == 1 && grp[1].second == 1 && grp[0].first < 2048 && grp[1].first < 2048)
            v = (1u << 29) | grp[0].first | (grp[1].first << 11);
        else {
            if (post.size() + 1 + 2 * grp.size() >= 0x3fffffffull) return fail(RK_ERR_LIMIT, "postings overflow"); // (offsets stay below 2^30: the k-mer-space value table uses the two top bits)
            v = 0x80000000u | (uint32_t)post.size();
            post.push_back((uint32_t)grp.size());
            // Order inside a list is free.  The fused kernels walk a list 16 postings per step and add to packed per-reference
            // counters, four (or two) references per LDS word: in ascending order the 16 lanes of a step meet four by four in one word
            // (the genomes of one family have consecutive ids) and the LDS serves them one after the other.  Ordered by
            // (ref mod 4, ref) a step's postings fall into 16 different words instead.
            static const bool spread = !(getenv("RKMH_POST_ORDER") && atoi(getenv("RKMH_POST_ORDER")) == 0);
            if (spread) std::stable_sort(grp.begin(), grp.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return (a.first & 3u) < (b.first & 3u); });
            for (auto& g : grp) { post.push_back(g.first); post.push_back(g.second); }
        }
        uint32_t b = index_bucket(pairs[i].h, bmask);
        for (;;) {
            uint32_t q = 0;
            while (q < (uint32_t)IDX_SLOTS && fpb[(size_t)IDX_SLOTS * b + q] != 0) ++q;
            if (q < (uint32_t)IDX_SLOTS) {
                const size_t sl = (size_t)IDX_SLOTS * b + q;
                fpb[sl] |= (uint16_t)index_fp(pairs[i].h);
                kv[4 * sl] = (uint32_t)pairs[i].h; kv[4 * sl + 1] = (uint32_t)(pairs[i].h >> 32); kv[4 * sl + 2] = v;
                break;
            }
            fpb[(size_t)IDX_SLOTS * b] |= (uint16_t)IDX_OVF; // the key goes further down the chain: lookups must follow
            b = (b + 1) & bmask;
        }
        i = j;
    }
    clk.tick("bucket table");
    RKCHK(c->d_fpb.reserve((size_t)size * 2));
    // compact the key/value entries: key id = (keys stored in earlier buckets) + position in the bucket
    std::vector<uint32_t> base((size_t)nb + 1, 0);
    for (uint32_t b = 0; b < nb; ++b) {
        uint32_t q = 0;
        while (q < (uint32_t)IDX_SLOTS && fpb[(size_t)IDX_SLOTS * b + q] != 0) ++q;
        base[(size_t)b + 1] = base[b] + q;
    }
    const size_t nkeys = base[nb];
    c->nkeys = (uint32_t)nkeys;
    ++c->index_gen;
    c->ix.keepkey = nullptr; memset(&c->ksets_m, 0, sizeof c->ksets_m); // a depth filter set earlier refers to the old key ids
    std::vector<uint32_t> dense((nkeys + 1) * 4, 0);
    for (uint32_t b = 0; b < nb; ++b)
        for (uint32_t q = 0; q < base[(size_t)b + 1] - base[b]; ++q)
            memcpy(&dense[((size_t)base[b] + q) * 4], &kv[((size_t)IDX_SLOTS * b + q) * 4], 16);
    c->h_keyhash.resize(nkeys);
    for (size_t q = 0; q < nkeys; ++q) c->h_keyhash[q] = ((uint64_t)dense[q * 4 + 1] << 32) | dense[q * 4];
    RKCHK(c->d_base.reserve(((size_t)nb + 1) * 4));
    HIPCHK(hipMemcpy(c->d_base.p, base.data(), ((size_t)nb + 1) * 4, hipMemcpyHostToDevice));
    RKCHK(c->d_kv.reserve((nkeys + 1) * 16));
    RKCHK(c->d_post.reserve(post.size() * 4 + 256)); // the k-mer-space kernel reads a hit's first sixteen postings before it knows the list's length
    HIPCHK(hipMemcpy(c->d_fpb.p, fpb.data(), (size_t)size * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_kv.p, dense.data(), (nkeys + 1) * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_post.p, post.data(), post.size() * 4, hipMemcpyHostToDevice));
    c->ix.fpb = c->d_fpb.as<uint4>(); c->ix.base = c->d_base.as<uint32_t>(); c->ix.kv = c->d_kv.as<uint4>();
    c->ix.post = c->d_post.as<uint32_t>();
    c->ix.bmask = bmask; c->ix.bshift = bshift; c->ix.nref = R;
    // First-level filter in front of the bucket table (RKMH_PREFILTER=0 turns it off for A/B runs): two bits set per key,
    // 32 bits per key where that fits in 1 MB -- measured at C2 (163 k keys): 256 KB 0.976 ms, 512 KB 0.959, 1 MB 0.953,
    // 2 MB 0.997 (the filter then crowds the reads and the table out of the 4 MB L2); at 10^6 keys 1 MB beats 2 MB (1.07 vs
    // 1.12 ms) although one window in twenty then passes by chance; only beyond 2 * 10^6 keys does 2 MB win (4 * 10^6
    // keys: 1 MB 2.28 ms, 2 MB 1.77, 4 MB 1.99).  RKMH_PRE_BITS / RKMH_PRE_MAXKB override both numbers.
    c->ix.pre = nullptr; c->ix.pmask = 0;
    int pre_mode = 1;
    if (const char* e = getenv("RKMH_PREFILTER")) pre_mode = atoi(e);
    if (pre_mode > 0) {
        size_t bits_per_key = 32, max_words = (size_t)(distinct > 2000000 ? 2048 : 1024) * 256;
        if (const char* e = getenv("RKMH_PRE_BITS")) { long v = atol(e); if (v >= 2 && v <= 256) bits_per_key = (size_t)v; }
        if (const char* e = getenv("RKMH_PRE_MAXKB")) { long v = atol(e); if (v >= 16 && v <= (1 << 20)) max_words = (size_t)v * 256; }
        uint32_t pwords = 1u << 12;
        while ((size_t)pwords * 32 < distinct * bits_per_key && (size_t)pwords * 2 <= max_words) pwords <<= 1;
        std::vector<uint32_t> pre(pwords, 0);
        for (size_t q = 0; q < pairs.size(); ++q) pre[index_pre_word(pairs[q].h, pwords - 1)] |= index_pre_bits(pairs[q].h);
        RKCHK(c->d_pre.reserve((size_t)pwords * 4));
        HIPCHK(hipMemcpy(c->d_pre.p, pre.data(), (size_t)pwords * 4, hipMemcpyHostToDevice));
        c->ix.pre = c->d_pre.as<uint32_t>(); c->ix.pmask = pwords - 1;
    }
    // k-mer-space structures (every k-mer size of the run from 8 to 16): every k-mer of the 4^k universe whose canonical hash is a key
    // (or 0), found by exhaustive enumeration on the device -- see k_enum_kmers -- goes into the group filter and the exact map of
    // k_classify_kmer (rk_kmer.hip), one pair per size.  RKMH_KMER_PREFILTER=0 turns them off (A/B runs, tests).
    c->ix.kpk = 0; c->kpre_inserted = 0;
    c->ix.kf4 = nullptr; c->ix.kf4_n = 0; c->ix.km1 = nullptr; c->ix.km1_b = 0; c->ix.km1_vals = nullptr;
    memset(&c->ksets, 0, sizeof c->ksets);
    static const int kmer_env = getenv("RKMH_KMER_PREFILTER") ? atoi(getenv("RKMH_KMER_PREFILTER")) : -1;
    const int kmer_mode = kmer_env >= 0 ? kmer_env : (pre_mode > 0 ? 1 : 0);
    static const long kmer_max_keys_env = getenv("RKMH_KPRE_MAXKEYS") ? atol(getenv("RKMH_KPRE_MAXKEYS")) : -1;
    const size_t kmer_max_keys = kmer_max_keys_env >= 0 ? (size_t)kmer_max_keys_env : 6000000;
    bool all_k_ok = kmer_mode > 0 && c->kmer_form_allowed && c->ks.n >= 1 && c->ks.n <= KM_MAX_KS && distinct <= kmer_max_keys;
    // one k of 17 .. 20 (wide k-mers, 64-bit): the 4^k enumeration takes 0.1 s (k = 17), 0.4 s (18), 1.7 s (19), 6.7 s (20) -- done unasked
    // up to RKMH_KMER_ENUM_MAXK (default 18); beyond that only when the cache file (rk_set_kmer_cache) already holds the list
    const int enum_maxk = getenv("RKMH_KMER_ENUM_MAXK") ? atoi(getenv("RKMH_KMER_ENUM_MAXK")) : 18; // (read per build: a few per process)
    const bool wide_k = c->ks.n == 1 && c->ks.k[0] > 16 && c->ks.k[0] <= KW_MAX_K;
    for (int j = 0; j < c->ks.n; ++j) all_k_ok = all_k_ok && c->ks.k[j] >= KPRE_MIN_K && (c->ks.k[j] <= 16 || wide_k);
    if (wide_k && distinct >= (size_t)KW_EMPTY - 16) all_k_ok = false; // (key numbers of the wide map are 20 bits)
    for (int j = 0; j + 1 < c->ks.n; ++j) for (int i = j + 1; i < c->ks.n; ++i) all_k_ok = all_k_ok && c->ks.k[j] != c->ks.k[i]; // a size given twice hashes twice: hash-space path
    clk.tick("index + prefilter uploaded");
    std::vector<uint32_t> kpost, kbase;
    std::unordered_map<uint32_t, KList> kremap;
    c->ix.kpost = nullptr; c->ix.kbase = nullptr; c->ix.kkeys = nullptr; c->ix.kslots = nullptr;
    if (all_k_ok) {
        build_kpost(post, R, kpost, kbase, kremap);
        if (kpost.size() >= 0x3fffffffull) all_k_ok = false;
        else {
            RKCHK(c->d_kpost.reserve(kpost.size() * 4));
            RKCHK(c->d_kbase.reserve(kbase.size() * 4));
            HIPCHK(hipMemcpy(c->d_kpost.p, kpost.data(), kpost.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c->d_kbase.p, kbase.data(), kbase.size() * 4, hipMemcpyHostToDevice));
            c->ix.kpost = c->d_kpost.as<uint32_t>(); c->ix.kbase = c->d_kbase.as<uint32_t>();
        }
    }
    clk.tick("posting lists (kpost)");
    std::map<int, std::vector<uint32_t>> kcache;
    bool kcache_dirty = false;
    uint64_t kcache_tag = 0;
    c->kmer_cache_state = 0;
    if (all_k_ok && !c->kmer_cache_path.empty()) {
        kcache_tag = kmer_cache_tag(c, dense, nkeys);
        if (kmer_cache_read(c->kmer_cache_path, kcache_tag, kcache)) c->kmer_cache_state = 1;
    }
    if (all_k_ok && wide_k && c->ks.k[0] > enum_maxk && kcache.find(c->ks.k[0]) == kcache.end()) all_k_ok = false; // too long to do unasked
    std::vector<uint8_t> seen(all_k_ok ? nkeys + 1 : 0, 0); // across the sizes: a key found by two k-mers of ANY sizes disables the form
    int built = 0;
    for (int kidx = 0; all_k_ok && kidx < c->ks.n; ++kidx) {
        const int k = c->ks.k[kidx];
        // the k-mers found come back as a list (one per strand pair): normally exactly one per key, plus any k-mer that collides
        // with a key or hashes to 0 -- a handful at most, so twice the keys is ample room; more than that disables this form
        const uint32_t list_cap = (uint32_t)std::min<size_t>(2 * distinct + 4096, 0x3fffffffu);
        DevBuf d_list, d_stats;
        struct Release { DevBuf& a; DevBuf& b; ~Release() { a.release(); b.release(); } } release_list{d_list, d_stats}; // freed on every path out
        uint32_t found = 0;
        std::vector<uint32_t> list;
        auto cached = kcache.find(k);
        const size_t lw = k > 16 ? 3 : 2; // words per list item on the host: (k-mer [low, high], key id)
        if (cached != kcache.end()) { // the enumeration of an earlier run with these keys, this k and this hashing policy
            list = cached->second;
            found = (uint32_t)(list.size() / lw);
        } else {
            RKCHK(d_list.reserve((size_t)list_cap * (k > 16 ? 16 : 8)));
            RKCHK(d_stats.reserve(16));
            HIPCHK(hipMemsetAsync(d_stats.p, 0, 16, c->st));
            hipError_t le = launch_enum_kmers(c->ix, c->pol, k, d_stats.as<uint32_t>(), d_list.as<uint2>(), list_cap, c->st);
            if (le == hipSuccess) le = hipMemcpyAsync(&found, d_stats.p, 4, hipMemcpyDeviceToHost, c->st);
            if (le == hipSuccess) le = hipStreamSynchronize(c->st);
            const size_t got = std::min<uint32_t>(found, list_cap);
            std::vector<uint32_t> raw(got * (k > 16 ? 4 : 2));
            if (le == hipSuccess && !raw.empty()) le = hipMemcpy(raw.data(), d_list.p, raw.size() * 4, hipMemcpyDeviceToHost);
            if (le != hipSuccess) return fail(RK_ERR_HIP, "k-mer enumeration: %s", hipGetErrorString(le));
            // (the device appends in a racy order: sorted by k-mer, the list -- and the cache file -- is reproducible)
            std::vector<std::pair<uint64_t, uint32_t>> items(got);
            for (size_t i = 0; i < got; ++i)
                items[i] = k > 16 ? std::make_pair(((uint64_t)raw[4 * i + 1] << 32) | raw[4 * i], raw[4 * i + 2]) : std::make_pair((uint64_t)raw[2 * i], raw[2 * i + 1]);
            std::sort(items.begin(), items.end());
            list.resize(got * lw);
            for (size_t i = 0; i < got; ++i) {
                list[lw * i] = (uint32_t)items[i].first;
                if (k > 16) list[lw * i + 1] = (uint32_t)(items[i].first >> 32);
                list[lw * i + lw - 1] = items[i].second;
            }
            if (found <= list_cap && !c->kmer_cache_path.empty()) { kcache[k] = list; kcache_dirty = true; }
        }
        c->kpre_inserted += found;
        // Built only when every key has exactly one preimage (found == keys + zero-hash k-mers with no two entries sharing a key id):
        // the k-mer then identifies the key in the per-read hit multiset.  Anything else leaves the hash-space kernels in charge.
        bool ok = found <= list_cap;
        if (getenv("RKMH_KMAP_FORCE_DUP")) ok = false; // tests: behave as if two k-mers shared a key (nothing is built)
        clk.tick(cached != kcache.end() ? "k-mer lists from the cache" : "k-mer enumeration");
        if (ok) {
            for (uint32_t i = 0; ok && i < found; ++i) {
                const uint32_t slot = list[lw * (size_t)i + lw - 1];
                if (slot == IDX_NOT_FOUND) continue;
                if (slot >= nkeys || seen[slot]) ok = false; // two different k-mers with the same 64-bit canonical hash
                else seen[slot] = 1;
            }
        }
        // value id of an index key for the k-mer-space kernels: the reference itself (one posting, once) or nref + the number of a
        // compound value of four dwords in `vals` (km1_vals) -- shared by the narrow and the wide form
        std::vector<uint32_t> vals;
        std::unordered_map<uint32_t, uint32_t> val_id;
        auto value_id_of = [&](uint32_t slot) -> uint32_t {
            const uint32_t val = dense[(size_t)slot * 4 + 2];
            if (!(val >> 31) && ((val >> 29) & 3u) == 0u && ((val >> 20) & 0x1FFu) == 1u) return val & 0xFFFFFu; // one posting, once: the reference
            // lists: identical ones share one compound value (and one copy in kpost, see build_kpost)
            uint32_t vkey = val;
            KList kl;
            if (val >> 31) { kl = kremap.at(val & 0x7fffffffu); vkey = 0x80000000u | kl.plain; }
            auto it = val_id.find(vkey);
            if (it == val_id.end()) {
                it = val_id.emplace(vkey, (uint32_t)(R + vals.size() / 4)).first;
                // four dwords per entry: the index value and, for a list of three to six references that each hold the hash
                // once (what related types of one panel share), the list itself, nine bits per reference -- the kernel then
                // counts it in the lane that found the hit instead of fetching the posting list from global memory (KM1V_INLINE)
                uint32_t x = val, y = 0;
                if (RK_KMER_INLINE_N && (val >> 31)) {
                    const uint32_t off = val & 0x7fffffffu, n = post[off];
                    bool ok3 = n >= 3 && n <= 6;
                    for (uint32_t q = 0; ok3 && q < n; ++q) ok3 = post[off + 1 + 2 * q] < 512u && post[off + 2 + 2 * q] == 1u;
                    if (ok3) {
                        uint32_t r[6] = {0, 0, 0, 0, 0, 0};
                        for (uint32_t q = 0; q < n; ++q) r[q] = post[off + 1 + 2 * q];
                        x = 0xC0000000u | ((n - 3u) << 27) | r[0] | (r[1] << 9) | (r[2] << 18);
                        y = r[3] | (r[4] << 9) | (r[5] << 18);
                    }
                }
                // a list that stays a list: x = the form the dense-counter kernels walk (plain, or base + exceptions) -- or, within eight
                // exceptions of its base, x, y and w hold base and exceptions themselves -- and z = the plain form (sparse counters)
                uint32_t z = 0, w = 0;
                if ((x >> 30) == 2u) {
                    z = kl.plain;
                    if (kl.ix) { x = kl.ix; y = kl.iy; w = kl.iw; } else x = 0x80000000u | kl.enc;
                }
                vals.push_back(x); vals.push_back(y); vals.push_back(z); vals.push_back(w);
            }
            return it->second;
        };
        if (ok && k > 16) {
            // ---- wide k-mers: the same group filter (sector and bits from kw_fold of core and k-mer), the km2 map and kkeys ----
            std::vector<uint64_t> km(found);
            for (uint32_t i = 0; i < found; ++i) km[i] = ((uint64_t)list[3 * (size_t)i + 1] << 32) | list[3 * (size_t)i];
            const double want = (double)found * 8.0 / (found > 300000u ? 18.0 : 13.0);
            const uint32_t nsect = want < 256.0 ? 256u : (want > 16777216.0 ? 16777216u : ((uint32_t)want + 7u) & ~7u);
            std::vector<uint32_t> f4((size_t)4 * nsect, 0u);
            const uint64_t cm = (1ull << (2 * (k - 3))) - 1ull;
            for (uint32_t i = 0; i < found; ++i) {
                const uint64_t v = km[i], rv = packed_revcomp64(v, k);
                for (int o = 0; o < (rv == v ? 1 : 2); ++o) {
                    const uint64_t X = o ? rv : v;
                    const uint32_t bits = kf4_bits(kw_fold(X));
                    for (uint32_t j = 0; j < 4; ++j) f4[(size_t)kf4_sector(kw_fold((X >> (2 * (3 - j))) & cm), nsect) * 4 + j] |= bits;
                }
            }
            std::vector<uint32_t> kk((size_t)found * 2 + 4, 0u), kslot((size_t)found + 4, 0u);
            for (uint32_t i = 0; ok && i < found; ++i) {
                const uint32_t slot = list[3 * (size_t)i + 2];
                const uint32_t vid_ = slot == IDX_NOT_FOUND ? KW_VID_ZERO : value_id_of(slot);
                if (vid_ >= KW_VID_ZERO && slot != IDX_NOT_FOUND) ok = false; // (value ids are 24 bits here)
                kk[2 * (size_t)i] = (uint32_t)km[i]; kk[2 * (size_t)i + 1] = (uint32_t)(km[i] >> 32) | (vid_ << 8);
                kslot[i] = slot == IDX_NOT_FOUND ? 0u : slot;
            }
            const uint32_t kbits = 2u * (uint32_t)k;
            uint32_t b = 12;
            static const double km2_load = getenv("RKMH_KM2_LOAD") ? atof(getenv("RKMH_KM2_LOAD")) : 0.65;
            while (b < 26 && (double)found > km2_load * 4.0 * (double)((size_t)1 << b)) ++b;
            std::vector<uint32_t> c1;
            bool placed_all = false;
            for (; b <= 26 && !placed_all; ++b) {
                const uint32_t nbk = 1u << b;
                c1.assign((size_t)nbk * 4, KW_EMPTY); // empty: key number all ones, hop / tag / flag clear
                placed_all = true;
                for (uint32_t i = 0; i < found && placed_all; ++i) {
                    const uint64_t y = kw_y(km[i], k);
                    uint32_t bk = (uint32_t)(y >> (kbits - b));
                    const uint32_t tag = (uint32_t)(y >> (kbits - b - KW_TAG)) & ((1u << KW_TAG) - 1u);
                    bool placed = false;
                    for (uint32_t hop = 0; hop < (1u << KM1_HB) && !placed; ++hop) {
                        uint32_t* e = &c1[(size_t)bk * 4];
                        for (int q = 0; q < 4 && !placed; ++q)
                            if ((e[q] & KW_EMPTY) == KW_EMPTY) {
                                e[q] = (e[q] & (1u << KW_IDBITS)) | (((hop << KW_TAG) | tag) << (KW_IDBITS + 1)) | i; // (the flag of a last cell stays)
                                placed = true;
                            }
                        if (!placed) { e[3] |= 1u << KW_IDBITS; bk = (bk + 1) & (nbk - 1); }
                    }
                    placed_all = placed;
                }
                if (placed_all) break;
            }
            if (!placed_all) ok = false;
            else {
                RKCHK(c->d_km1[(size_t)kidx].reserve(c1.size() * 4));
                HIPCHK(hipMemcpy(c->d_km1[(size_t)kidx].p, c1.data(), c1.size() * 4, hipMemcpyHostToDevice));
                RKCHK(c->d_km1v[(size_t)kidx].reserve(vals.size() * 4 + 16));
                if (!vals.empty()) HIPCHK(hipMemcpy(c->d_km1v[(size_t)kidx].p, vals.data(), vals.size() * 4, hipMemcpyHostToDevice));
                RKCHK(c->d_kkeys.reserve(kk.size() * 4));
                HIPCHK(hipMemcpy(c->d_kkeys.p, kk.data(), kk.size() * 4, hipMemcpyHostToDevice));
                RKCHK(c->d_kslots.reserve(kslot.size() * 4));
                HIPCHK(hipMemcpy(c->d_kslots.p, kslot.data(), kslot.size() * 4, hipMemcpyHostToDevice));
                RKCHK(c->d_kf4[(size_t)kidx].reserve(f4.size() * 4));
                HIPCHK(hipMemcpy(c->d_kf4[(size_t)kidx].p, f4.data(), f4.size() * 4, hipMemcpyHostToDevice));
                c->ksets.km1[kidx] = c->d_km1[(size_t)kidx].as<uint4>(); c->ksets.km1_b[kidx] = b; c->ksets.km1_vals[kidx] = c->d_km1v[(size_t)kidx].as<uint32_t>();
                c->ksets.kf4[kidx] = c->d_kf4[(size_t)kidx].as<uint4>(); c->ksets.kf4_n[kidx] = nsect; c->ksets.k[kidx] = k;
                c->ix.kkeys = c->d_kkeys.as<uint2>(); c->ix.kslots = c->d_kslots.as<uint32_t>();
                c->km1_ncells[kidx] = 0;
                ++built;
            }
            clk.tick("wide filter + map");
            if (!ok) break;
            continue;
        }
        if (ok) {
            // group filter of k_classify_kmer (kf4_sector in rk_device.hpp): every found k-mer in both orientations under its four
            // alignments, RK_KF4_NBITS (three) bits each in dword j of the 16-byte sector its alignment-j core selects (at 14 entries per
            // sector about 9 of a dword's 32 bits are set: one window in ~45 of those that hit nothing passes by chance)
            // Size (any sector count, kf4_sector scales the hashed core): a sparser filter sends fewer windows to the exact map, a
            // smaller one leaves more of an XCD's 4 MB of L2 to the map and the streaming bases -- and the second matters more until
            // the panel is far beyond any cache.  Measured optimum, entries per sector (tools/kf4_density.sh, 1 M reads; ms at the
            // optimum / at the 5-10 a power-of-two size would give): 161 k keys (C2, 1 MB map) 12-13.5 (0.321 / 0.335); 239 k keys
            // (266 references, C3; 2 MB map) 12.5-16 (0.343 / 0.425); 270 k 14 (0.353 / 0.433); 360 k (4 MB map) 20 (0.397 / 0.584);
            // 540 k 20-24 (0.592 / 0.655); 900 k (8 MB map) 14 (0.747 / 0.787); 1.8 M <= 10 (0.894); 3.6 M <= 10 (0.947).
            uint32_t nsect = 0;
            {
                static const double kf4_entries = getenv("RKMH_KF4_ENTRIES") ? atof(getenv("RKMH_KF4_ENTRIES")) : 0.0; // forced density (A/B runs)
                static const double km1_load_est = getenv("RKMH_KM1_LOAD") ? atof(getenv("RKMH_KM1_LOAD")) : 0.65;
                uint32_t be = 2u * (uint32_t)k < 12u ? 2u * (uint32_t)k : 12u;           // the map's size, as its builder below will choose it
                while (be < 2u * (uint32_t)k && be < 28 && (double)found > km1_load_est * 4.0 * (double)((size_t)1 << be)) ++be;
                const size_t map_bytes = (size_t)16 << be;
                // (k = 16 with s = 2000, 322 k keys: 20 entries 0.646, 13 entries 0.665; k = 12, whose 9-base cores crowd the sectors
                // unevenly: 6-8 entries 0.477, 10 entries 0.504, 13 entries 0.555; k = 13: 9-13 entries 0.40, 6 entries 0.435)
                // (all of the above with two bits per entry; with the three shipped -- kf4_bits -- the optima move little: C2 14 entries
                // 0.314, 12 0.317, 17 0.326; 266 references 13-14 0.332; s = 2000 16 0.626, 20 0.635; 400 references 20 0.407)
                // (k = 15 / 14 with three bits: 12.5 entries 0.348 / 0.366, 14 entries 0.360 / 0.369, 11 entries 0.353 / 0.375)
                double e = k <= 12 ? 7.0 : (k == 13 ? 10.0 : 13.0);
                if (found > 1500000u) e = 8.0;                                             // far beyond any cache: fewer false candidates win
                else if (found > 300000u && map_bytes <= ((size_t)4 << 20)) e = 18.0;      // map and filter fight for the L2: smallest useful filter
                if (kf4_entries > 0.0) e = kf4_entries;
                const double want = (double)found * 8.0 / e;
                nsect = want < 256.0 ? 256u : (want > 16777216.0 ? 16777216u : ((uint32_t)want + 7u) & ~7u);
            }
            std::vector<uint32_t> f4((size_t)4 * nsect, 0u);
            const uint32_t cm = kf4_core_mask(k);
            for (uint32_t i = 0; i < found; ++i) {
                const uint32_t v = list[2 * (size_t)i], rv = packed_revcomp(v, k);
                for (int o = 0; o < (rv == v ? 1 : 2); ++o) {
                    const uint32_t X = o ? rv : v, bits = kf4_bits(X);
                    for (uint32_t j = 0; j < 4; ++j)
                        f4[(size_t)kf4_sector((X >> (2 * (3 - j))) & cm, nsect) * 4 + j] |= bits;
                }
            }
            // exact map (KM1_C in rk_device.hpp).  A key whose bucket is full moves on by up to 2^KM1_HB - 1 buckets; if that is not
            // enough, or the value ids do not fit the cell, the table doubles (shorter remainders leave more bits for the id).
            {
                static const double km1_load = getenv("RKMH_KM1_LOAD") ? atof(getenv("RKMH_KM1_LOAD")) : 0.65;
                std::vector<uint32_t> vid(found);
                const uint32_t VID_ZERO = 0xFFFFFFFEu; // placeholder, mapped to the layout's id below
                for (uint32_t i = 0; i < found; ++i) {
                    const uint32_t slot = list[2 * (size_t)i + 1];
                    vid[i] = slot == IDX_NOT_FOUND ? VID_ZERO : value_id_of(slot);
                }
                const uint32_t kbits = 2u * (uint32_t)k;
                uint32_t b = kbits < 12u ? kbits : 12u;
                while (b < kbits && b < 28 && (double)found > km1_load * 4.0 * (double)((size_t)1 << b)) ++b;
                std::vector<uint32_t> c1;
                std::vector<uint32_t> cell_of(found); // where each found k-mer was placed (rk_set_depth_filter masks cells by key)
                bool built = false;
                for (; b <= kbits && b <= 28 && !built; ++b) {
                    const uint32_t r = kbits - b, vb = km1_vbits(k, b), vmask = (1u << vb) - 1u;
                    if ((uint64_t)R + vals.size() / 4 + 2 > (uint64_t)vmask) continue; // ids need more bits: a longer bucket index frees them
                    const uint32_t nbk = 1u << b, rmask = r ? (1u << r) - 1u : 0u;
                    c1.assign((size_t)nbk * 4, ~(1u << vb)); // empty: tag and id all ones, flag clear
                    bool placed_all = true;
                    for (uint32_t i = 0; i < found && placed_all; ++i) {
                        const uint32_t y = km1_y(list[2 * (size_t)i], k);
                        uint32_t bk = r ? y >> r : y;
                        const uint32_t rem = y & rmask, id = vid[i] == VID_ZERO ? vmask - 1u : vid[i];
                        bool placed = false;
                        for (uint32_t hop = 0; hop < (1u << KM1_HB) && !placed; ++hop) {
                            uint32_t* e = &c1[(size_t)bk * 4];
                            for (int q = 0; q < 4 && !placed; ++q)
                                if ((e[q] & vmask) == vmask) { // empty (no key carries the all-ones id)
                                    e[q] = ((rem | (hop << r)) << (vb + 1)) | id;
                                    cell_of[i] = bk * 4u + (uint32_t)q;
                                    placed = true;
                                }
                            if (!placed) { e[3] |= 1u << vb; bk = (bk + 1) & (nbk - 1); } // full: later lookups that miss here try the next bucket
                        }
                        placed_all = placed;
                    }
                    if (placed_all) { built = true; break; }
                }
                if (built) { // else: the hash-space kernels serve the panel
                    DevBuf& d_km1 = c->d_km1[(size_t)kidx];
                    DevBuf& d_km1v = c->d_km1v[(size_t)kidx];
                    RKCHK(d_km1.reserve(c1.size() * 4));
                    HIPCHK(hipMemcpy(d_km1.p, c1.data(), c1.size() * 4, hipMemcpyHostToDevice));
                    RKCHK(d_km1v.reserve(vals.size() * 4 + 16));
                    if (!vals.empty()) HIPCHK(hipMemcpy(d_km1v.p, vals.data(), vals.size() * 4, hipMemcpyHostToDevice));
                    c->ksets.km1[kidx] = d_km1.as<uint4>(); c->ksets.km1_b[kidx] = b; c->ksets.km1_vals[kidx] = d_km1v.as<uint32_t>();
                    std::vector<uint32_t> cells(2 * (size_t)found);
                    for (uint32_t i = 0; i < found; ++i) { cells[2 * (size_t)i] = cell_of[i]; cells[2 * (size_t)i + 1] = list[2 * (size_t)i + 1]; }
                    RKCHK(c->d_km1cells[(size_t)kidx].reserve(cells.size() * 4 + 16));
                    if (found) HIPCHK(hipMemcpy(c->d_km1cells[(size_t)kidx].p, cells.data(), cells.size() * 4, hipMemcpyHostToDevice));
                    c->km1_ncells[kidx] = found; c->km1_vmask[kidx] = (1u << km1_vbits(k, b)) - 1u;
                }
            }
            DevBuf& d_kf4 = c->d_kf4[(size_t)kidx];
            RKCHK(d_kf4.reserve(f4.size() * 4));
            HIPCHK(hipMemcpy(d_kf4.p, f4.data(), f4.size() * 4, hipMemcpyHostToDevice));
            if (c->ksets.km1[kidx]) { c->ksets.kf4[kidx] = d_kf4.as<uint4>(); c->ksets.kf4_n[kidx] = nsect; c->ksets.k[kidx] = k; ++built; }
            else ok = false;
        }
        if (!ok) break; // one size without its structures: the hash-space kernels serve the run
    }
    clk.tick("filter + exact map");
    if (kcache_dirty) c->kmer_cache_state = kmer_cache_write(c->kmer_cache_path, kcache_tag, kcache) ? 2 : 3;
    if (built == c->ks.n && built > 0) { // every size has its filter and map
        c->ksets.n = built;
        c->ix.kf4 = c->ksets.kf4[0]; c->ix.kf4_n = c->ksets.kf4_n[0]; c->ix.km1 = c->ksets.km1[0]; c->ix.km1_b = c->ksets.km1_b[0];
        c->ix.km1_vals = c->ksets.km1_vals[0]; c->ix.kpk = (uint32_t)c->ksets.k[0];
    } else memset(&c->ksets, 0, sizeof c->ksets);
    // a full bottom-S sketch of uniform hashes keeps the fraction (largest kept hash / 2^64) of the k-mers
    c->density = 0.0;
    for (int r = 0; r < R; ++r) {
        const int len = c->h_lens[(size_t)r];
        double d = 1.0;
        if (len == S && len > 0) d = (double)c->h_sk[(size_t)r * S + (size_t)len - 1] / 18446744073709551616.0;
        if (d > c->density) c->density = d;
    }
    c->have_refs = true;
    return build_key_mask(c); // a bounded depth filter set earlier follows the new key ids
}

extern "C" int rk_set_reference_sketches(rk_ctx* c, const uint64_t* sketches, const int32_t* lens, int nref,
                                         const int* ks, int nks, int S) {
    if (!c || !sketches || !lens || nref < 1) return fail(RK_ERR_ARG, "bad arguments (need >= 1 reference)");
    if (S < 1 || S > RK_MAX_SKETCH) return fail(RK_ERR_LIMIT, "sketch size %d outside [1,%d]", S, RK_MAX_SKETCH);
    RKCHK(set_dev(c));
    RKCHK(check_ks(ks, nks, &c->ks));
    c->nref = nref; c->S = S;
    c->h_sk.assign(sketches, sketches + (size_t)nref * S);
    c->h_lens.assign(lens, lens + nref);
    for (int r = 0; r < nref; ++r)
        if (lens[r] < 0 || lens[r] > S) return fail(RK_ERR_ARG, "sketch length %d of reference %d outside [0,%d]", lens[r], r, S);
    return build_index(c);
}

// bases on the host, or (d_bases != nullptr) already on this context's device
static int set_references_impl(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases, const uint64_t* offsets, int nref,
                               const int* ks, int nks, int S, int max_samples, uint64_t counter_slots) {
    if (!c || !offsets || nref < 1) return fail(RK_ERR_ARG, "bad arguments (need >= 1 reference; rkmh.cpp:848 is undefined for 0)");
    if (S < 1 || S > RK_MAX_SKETCH) return fail(RK_ERR_LIMIT, "sketch size %d outside [1,%d]", S, RK_MAX_SKETCH);
    GeneralCfg cfg;
    RKCHK(check_ks(ks, nks, &cfg.ks));
    cfg.S = S;
    std::vector<uint64_t> sk((size_t)nref * S);
    std::vector<int32_t> lens((size_t)nref);
    GeneralOut go; go.sketches = sk.data(); go.lens = lens.data();
    rk_counter* cnt = nullptr;
    if (max_samples >= 0) {
        // -I path (rkmh.cpp:828-838): pass 1 counts every k-mer occurrence, pass 2 sketches with the range filter
        RKCHK(rk_counter_create(c, counter_slots ? counter_slots : 200000000ull, &cnt));
        GeneralCfg c1 = cfg;
        if (c->ref_count_mode == 1) c1.distinct_counter = cnt; else c1.inc_counter = cnt;
        GeneralOut none;
        int r = general_run(c, bases, d_bases, offsets, nref, c1, none);
        if (r != RK_OK) { rk_counter_destroy(cnt); return r; }
        cfg.filt_counter = cnt; cfg.filter_mode = FILTER_RANGE; cfg.fmin = 0; cfg.fmax = max_samples;
    }
    IndexClock clk;
    int r = general_run(c, bases, d_bases, offsets, nref, cfg, go);
    clk.tick("reference sketches (device)");
    if (cnt) rk_counter_destroy(cnt);
    if (r != RK_OK) return r;
    return rk_set_reference_sketches(c, sk.data(), lens.data(), nref, ks, nks, S);
}
extern "C" int rk_set_references(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int nref,
                                 const int* ks, int nks, int S, int max_samples, uint64_t counter_slots) {
    return set_references_impl(c, bases, nullptr, offsets, nref, ks, nks, S, max_samples, counter_slots);
}

extern "C" int rk_get_reference_sketches(rk_ctx* c, uint64_t* sketches, int32_t* lens) {
    if (!c || !sketches || !lens) return fail(RK_ERR_ARG, "bad arguments");
    if (!c->have_refs) return fail(RK_ERR_STATE, "no references set");
    memcpy(sketches, c->h_sk.data(), c->h_sk.size() * 8);
    memcpy(lens, c->h_lens.data(), c->h_lens.size() * 4);
    return RK_OK;
}
extern "C" int rk_num_references(const rk_ctx* c) { return c ? c->nref : 0; }

extern "C" int rk_set_reference_count_mode(rk_ctx* c, int mode) {
    if (!c || (mode != 0 && mode != 1)) return fail(RK_ERR_ARG, "mode must be 0 or 1");
    c->ref_count_mode = mode;
    return RK_OK;
}

extern "C" int rk_set_kmer_form(rk_ctx* c, int enable) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    c->kmer_form_allowed = enable != 0;
    return RK_OK;
}
extern "C" int rk_set_kmer_cache(rk_ctx* c, const char* path) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    c->kmer_cache_path = path ? path : "";
    return RK_OK;
}
extern "C" int rk_kmer_cache_state(const rk_ctx* c) { return c ? c->kmer_cache_state : 0; }
extern "C" int rk_kmer_form(const rk_ctx* c, uint32_t* kmers_found) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    if (!c->have_refs) return fail(RK_ERR_STATE, "no references set");
    if (kmers_found) *kmers_found = c->kpre_inserted;
    return c->ksets.n >= 1 ? 1 : 0;
}

// the per-key form of the depth filter (bounded min_num): keep bit per key id, masked copies of the exact k-mer maps
static int build_key_mask(rk_ctx* c) {
    c->ix.keepkey = nullptr;
    memset(&c->ksets_m, 0, sizeof c->ksets_m);
    if (!c->depth || c->min_num_bound < 0 || !c->have_refs) return RK_OK;
    RKCHK(set_dev(c));
    RKCHK(c->d_keepkey.reserve(((size_t)c->nkeys + 31) / 32 * 4 + 16));
    if (c->depth->compact && c->depth->index_gen != c->index_gen)
        return fail(RK_ERR_STATE, "the compact depth map was laid out for another reference set");
    HIPCHK(launch_keep_keys(c->ix, c->nkeys, c->depth->d, c->depth->slots, c->depth->compact ? c->depth->c_keysid.as<uint32_t>() : nullptr,
                            c->min_occ, c->pol, c->d_keepkey.as<uint32_t>(), c->st));
    if (c->ksets.n >= 1) {
        c->ksets_m = c->ksets;
        for (int j = 0; j < c->ksets.n; ++j) {
            if (c->ksets.k[j] > 16) continue; // wide k-mers: the kernel tests the key's keep bit itself (kkeys carries the key id)
            const size_t bytes = (size_t)16 << c->ksets.km1_b[j];
            RKCHK(c->d_km1m[(size_t)j].reserve(bytes));
            HIPCHK(hipMemcpyAsync(c->d_km1m[(size_t)j].p, c->ksets.km1[j], bytes, hipMemcpyDeviceToDevice, c->st));
            HIPCHK(launch_km1_mask(c->d_km1cells[(size_t)j].as<uint2>(), c->km1_ncells[j], c->d_keepkey.as<uint32_t>(),
                                   c->d_km1m[(size_t)j].as<uint32_t>(), c->km1_vmask[j], c->st));
            c->ksets_m.km1[j] = c->d_km1m[(size_t)j].as<uint4>();
        }
    }
    // the hash-space kernels: a copy of the key array with the verdict in each entry's fourth dword
    RKCHK(c->d_kvm.reserve(((size_t)c->nkeys + 1) * 16));
    HIPCHK(launch_kv_mask(c->ix.kv, c->nkeys, c->d_keepkey.as<uint32_t>(), c->d_kvm.as<uint4>(), c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    c->ix.keepkey = c->d_keepkey.as<uint32_t>();
    return RK_OK;
}

extern "C" int rk_set_depth_filter(rk_ctx* c, rk_counter* counter, int min_kmer_occ) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    if (counter && counter->compact && c->min_num_bound != 0)
        return fail(RK_ERR_STATE, "a compact depth map only answers min_num bound 0 (rk_set_min_num_bound(ctx, 0) first)");
    c->depth = nullptr; c->min_occ = min_kmer_occ;
    c->ix.keepkey = nullptr;
    memset(&c->ksets_m, 0, sizeof c->ksets_m);
    if (counter) {
        if (counter->compact && (counter->index_gen != c->index_gen || counter->ctx != c))
            return fail(RK_ERR_STATE, "the compact depth map was laid out for another reference set or context");
        c->depth = counter; // (every failure below leaves the context without a filter)
        struct Undo { rk_ctx* c; bool armed = true; ~Undo() { if (armed) { c->depth = nullptr; c->ix.keepkey = nullptr; } } } undo{c};
        // the fused kernel's masked forms read one KEEP bit per slot instead of the 4-byte count (k_keep_bits): a snapshot of
        // the table as it is NOW -- the -M flow sets the filter after pass 1 (and after the all-reduce in multi-GPU runs)
        RKCHK(set_dev(c));
        // pass 1 (rk_count_batch_device) is asynchronous on the CALLER's stream, an all-reduce may run on yet another one: the
        // snapshot must see the finished table, so the whole device is drained first (once per -M run: not a hot path)
        HIPCHK(hipDeviceSynchronize());
        RKCHK(counter_settle(counter));
        if (c->min_num_bound != 0) { // bound 0: no window is ever looked up by slot (the mask acts through the keys alone)
            RKCHK(c->d_keepbits.reserve(((counter->slots + 31) / 32) * 4 + 16));
            HIPCHK(launch_keep_bits(counter->d, counter->slots, min_kmer_occ, c->pol, c->d_keepbits.as<uint32_t>(), c->st));
            HIPCHK(hipStreamSynchronize(c->st));
        }
        RKCHK(build_key_mask(c));
        undo.armed = false;
    }
    return RK_OK;
}

// How much of min_num (row field 3) the caller needs under a depth filter.  num_mins only ever meets `num_mins <= min_matches`
// (src/rkmh.cpp:938; filter: `read_min_lens <= 0`, :1292), so a caller that compares with n needs min(min_num, n + 1) and no more.
extern "C" int rk_set_min_num_bound(rk_ctx* c, int bound) {
    if (!c) return fail(RK_ERR_ARG, "ctx is NULL");
    const int nb = bound < 0 ? -1 : bound;
    if (nb == c->min_num_bound) return RK_OK;
    if (c->depth && c->depth->compact && nb != 0) return fail(RK_ERR_STATE, "the depth filter in use is a compact map: it only answers min_num bound 0");
    c->min_num_bound = nb;
    if (c->depth) return rk_set_depth_filter(c, c->depth, c->min_occ); // rebuild the snapshot in the other form
    return RK_OK;
}
extern "C" int rk_min_num_bound(const rk_ctx* c) { return c ? c->min_num_bound : -1; }

// ---- the hot loop -------------------------------------------------------------------------------
// The -M count pass in its slot-partitioned form (rk_count.hip): worth its fixed cost (six launches, two passes over a slot
// array) for batches of millions of windows into tables that do not fit a few workgroups' LDS; RKMH_COUNT_BINS=1 / 0 forces it
// on (any size: the tests) / off (one device atomic per window, 2.6e10/s)
static int count_bins_env() {
    const char* e = getenv("RKMH_COUNT_BINS"); // read per pass (a few launches each): tests switch it inside one process
    return e && *e ? atoi(e) : -1;
}
static int count_partitioned(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, uint32_t ml, int expect,
                             rk_counter* k, uint64_t total_bases, hipStream_t st, bool* done) {
    *done = false;
    const int env = count_bins_env();
    if (env == 0 || total_bases == 0 || !classify_tile_supported(0, (int)ml)) return RK_OK;
    const uint64_t stride = (total_bases + 3) & ~3ull;
    CountPlan pl;
    if (stride >= (1ull << 31) || !count_plan(k->slots, stride * (uint64_t)c->ks.n, &pl)) return RK_OK;
    if (env < 0 && (pl.n < ((uint64_t)4 << 20) || pl.nsub < 256)) return RK_OK;
    std::lock_guard<std::mutex> lock(k->mu);
    if (!k->last) HIPCHK(hipEventCreateWithFlags(&k->last, hipEventDisableTiming));
    if (k->last_set) HIPCHK(hipStreamWaitEvent(st, k->last, 0)); // the previous pass into this table: scratch and sub-ranges are its
    if (k->last_atomic_set) HIPCHK(hipStreamWaitEvent(st, k->last_atomic, 0)); // atomics still landing would race with the plain adds
    const size_t need = count_plan_scratch_bytes(pl);
    if (need > k->ws.cap) { HIPCHK(hipDeviceSynchronize()); RKCHK(k->ws.reserve(need)); } // nothing may still be reading the old arrays
    const CountScratch sc = count_plan_carve(pl, k->ws.p);
    HIPCHK(launch_count_prepare(pl, sc, st));
    HIPCHK(launch_classify_tile((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, c->ks, c->S, c->ix, k->d, k->slots, 0, 1,
                                (int32_t*)sc.flat, c->pol, (int)ml, expect, st, (uint32_t)stride));
    HIPCHK(launch_count_bins(pl, sc, k->d, st));
    HIPCHK(hipEventRecord(k->last, st));
    k->last_set = true;
    *done = true;
    return RK_OK;
}

static int fused_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, void* d_out4,
                        uint32_t max_read_len, int mode, rk_counter* count_into, hipStream_t st, uint64_t total_bases = 0) {
    if (nreads > 0xfffffff0ll) return fail(RK_ERR_LIMIT, "more than 2^32-16 reads in one device batch");
    if (((uintptr_t)d_bases & 3) != 0) return fail(RK_ERR_ARG, "d_bases must be 4-byte aligned");
    int32_t* counter = nullptr; uint64_t slots = 1; int min_occ = 0;
    const bool bounded = mode != 1 && c->depth && c->min_num_bound >= 0; // the mask acts per key: no slot bitmap in the kernels
    if (bounded && !c->ix.keepkey) return fail(RK_ERR_STATE, "depth filter: the per-key mask was not built");
    if (mode == 1) { counter = count_into->d; slots = count_into->slots; }
    else if (c->depth && !bounded) { counter = c->d_keepbits.as<int32_t>(); slots = c->depth->slots; min_occ = c->min_occ; } // the keep bitmap, see rk_set_depth_filter
    uint32_t ml = max_read_len < 1 ? 1 : (max_read_len > (uint32_t)FUSED_MAXLEN ? (uint32_t)FUSED_MAXLEN : max_read_len);
    int expect = 0; // hits an error-free read is expected to score: sizes the kernel's per-read hit multiset
    for (int j = 0; j < c->ks.n; ++j) expect += (int)(c->density * (double)num_windows((int)ml, c->ks.k[j], c->pol.drop_last_window)) + 1;
    if (mode == 1 && count_into->compact) {
        // pass 1 into a compact depth map: hash every window, count the few whose slot is tracked (k_classify_tile, MODE 1, cs.tab)
        if (count_into->index_gen != c->index_gen || count_into->ctx != c)
            return fail(RK_ERR_STATE, "the compact depth map was laid out for another reference set or context");
        uint64_t nh = 0;
        for (int j = 0; j < c->ks.n; ++j) nh += (uint64_t)num_windows((int)max_read_len, c->ks.k[j], c->pol.drop_last_window);
        if (nh > (uint64_t)c->S || max_read_len > (uint32_t)FUSED_MAXLEN || !classify_tile_supported(0, (int)ml))
            return fail(RK_ERR_NEED_FULL, "reads of up to %u bases have more hashes (%llu) than the sketch keeps (%d): bottom-s selection needs the "
                        "depth of every hash, which a compact depth map does not hold", max_read_len, (unsigned long long)nh, c->S);
        RefIndex ix0 = c->ix; ix0.keepkey = nullptr;
        HIPCHK(launch_classify_tile((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, c->ks, c->S, ix0,
                                    counter, slots, 0, 1, nullptr, c->pol, (int)ml, expect, st, 0, 0x7fffffff, &count_into->cs));
        std::lock_guard<std::mutex> lock(count_into->mu);
        if (!count_into->last_atomic) HIPCHK(hipEventCreateWithFlags(&count_into->last_atomic, hipEventDisableTiming));
        HIPCHK(hipEventRecord(count_into->last_atomic, st)); // readers of the map wait for the latest pass (they all add with atomics: no order among them)
        count_into->last_atomic_set = true;
        return RK_OK;
    }
    if (mode == 1) {
        bool done = false;
        RKCHK(count_partitioned(c, d_bases, d_offs, nreads, ml, expect, count_into, total_bases, st, &done));
        if (done) return RK_OK;
        // atomic form: other passes into this table may still be adding with plain stores
        std::lock_guard<std::mutex> lock(count_into->mu);
        if (count_into->last_set) HIPCHK(hipStreamWaitEvent(st, count_into->last, 0));
        // atomic passes are CHAINED too (each waits for the one before): last_atomic is a single event re-recorded by every pass, so
        // it only covers all of them if every pass already contains its predecessors -- otherwise a slot-partitioned pass that follows
        // two atomic passes on different streams would wait for the second one only and its plain adds could lose the first one's counts
        if (count_into->last_atomic_set) HIPCHK(hipStreamWaitEvent(st, count_into->last_atomic, 0));
        if (!classify_tile_supported(0, (int)ml)) return fail(RK_ERR_LIMIT, "count pass: batch not supported by the fused kernel");
        RefIndex ix0 = c->ix; ix0.keepkey = nullptr;
        HIPCHK(launch_classify_tile((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, c->ks, c->S, ix0,
                                    counter, slots, min_occ, 1, nullptr, c->pol, (int)ml, expect, st)); // (given an array there, it would write slots to it)
        // a later pass of either form must not overlap this one
        if (!count_into->last_atomic) HIPCHK(hipEventCreateWithFlags(&count_into->last_atomic, hipEventDisableTiming));
        HIPCHK(hipEventRecord(count_into->last_atomic, st));
        count_into->last_atomic_set = true;
        return RK_OK;
    }
    RefIndex ix = c->ix;
    if (!bounded) ix.keepkey = nullptr;
    else {
        if (c->ksets_m.n >= 1) ix.km1 = c->ksets_m.km1[0]; // (the compile-time-k kernels read the first size's structures from ix)
        ix.kv = c->d_kvm.as<uint4>();                       // (hash-space kernels: the key array with the mask's verdict in it)
    }
    const int nmin_cap = bounded ? c->min_num_bound : 0x7fffffff;
    // classification with k-mer sizes the exact k-mer maps were enumerated for: the k-mer-space kernel (rk_kmer.hip); under a
    // bounded depth filter it reads the masked copies of the maps (a dropped key is a zero-hash k-mer there)
    if (!counter && c->ksets.n == c->ks.n && c->ksets.n >= 1 && (!bounded || c->ksets_m.n == c->ksets.n) &&
        classify_kmer_supported(c->ix.nref, (int)ml, c->ks.k[0]))
        HIPCHK(launch_classify_kmer((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, bounded ? c->ksets_m : c->ksets, c->S, ix,
                                    (int32_t*)d_out4, c->pol, (int)ml, expect, st, nmin_cap));
    else if (classify_tile_supported(c->ix.nref, (int)ml))
        HIPCHK(launch_classify_tile((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, c->ks, c->S, ix,
                                    counter, slots, min_occ, 0, (int32_t*)d_out4, c->pol, (int)ml, expect, st, 0, nmin_cap));
    else
        HIPCHK(launch_fill_reroute((int32_t*)d_out4, (uint32_t)nreads, st)); // e.g. more than 16384 references: general path
    // bound > 0: the first `bound` surviving windows of every answered read are counted by hashing them (k_min_num_probe)
    if (bounded && c->min_num_bound > 0)
        HIPCHK(launch_min_num_probe((const uint8_t*)d_bases, (const uint32_t*)d_offs, (uint32_t)nreads, c->ks, c->S, c->min_num_bound,
                                    c->d_keepbits.as<uint32_t>(), c->depth->slots, c->pol, (int32_t*)d_out4, st));
    return RK_OK;
}

static int device_max_len(rk_ctx* c, const void* d_offs, int64_t nreads, hipStream_t st, uint32_t* out, uint32_t* end_off = nullptr) {
    RKCHK(c->w_misc.reserve(16));
    HIPCHK(launch_max_len((const uint32_t*)d_offs, (uint32_t)nreads, c->w_misc.as<uint32_t>(), st));
    HIPCHK(hipMemcpyAsync(out, c->w_misc.p, 4, hipMemcpyDeviceToHost, st));
    if (end_off) HIPCHK(hipMemcpyAsync(end_off, (const uint32_t*)d_offs + nreads, 4, hipMemcpyDeviceToHost, st)); // one past the last base
    HIPCHK(hipStreamSynchronize(st));
    return RK_OK;
}

// reroute reads the fused kernel flagged (max_id == -2) through the general path; offsets = u64 host offsets
static int reroute_flagged(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int32_t* out4) {
    std::vector<int64_t> idx;
    for (int64_t i = 0; i < nreads; ++i) if (out4[i * 4] == -2) idx.push_back(i);
    if (idx.empty()) return RK_OK;
    std::vector<uint64_t> offs(idx.size() + 1, 0);
    for (size_t j = 0; j < idx.size(); ++j) offs[j + 1] = offs[j] + (offsets[idx[j] + 1] - offsets[idx[j]]);
    std::vector<uint8_t> sub((size_t)offs.back() + 8);
    for (size_t j = 0; j < idx.size(); ++j) memcpy(sub.data() + offs[j], bases + offsets[idx[j]], (size_t)(offs[j + 1] - offs[j]));
    std::vector<int32_t> res(idx.size() * 4);
    GeneralCfg cfg; cfg.ks = c->ks; cfg.S = c->S; cfg.classify = true;
    apply_depth_cfg(c, cfg);
    GeneralOut go; go.out4 = res.data();
    RKCHK(general_run(c, sub.data(), nullptr, offs.data(), (int64_t)idx.size(), cfg, go));
    for (size_t j = 0; j < idx.size(); ++j) memcpy(out4 + idx[j] * 4, res.data() + j * 4, 16);
    return RK_OK;
}

// hpv16's per-read loop (src/rkmh.cpp:2656-2719): every hash of the read takes part (calc_hashes + mask + sort, no bottom-s);
// argmax over the first argmax_refs references (the HPV types, :2669-2679), raw intersection sizes against the others (the
// lineage- and sublineage-specific k-mer sets that sort_by_similarity ranks, :2688-2704).
extern "C" int rk_classify_groups_batch(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int argmax_refs,
                                        int32_t* out4, int32_t* tail_counts) {
    if (!c || !offsets || nreads < 0 || (nreads > 0 && (!out4 || !bases))) return fail(RK_ERR_ARG, "bad arguments");
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    if (argmax_refs < 1 || argmax_refs > c->ix.nref) return fail(RK_ERR_ARG, "argmax_refs %d outside [1,%d]", argmax_refs, c->ix.nref);
    if (argmax_refs < c->ix.nref && !tail_counts && nreads > 0) return fail(RK_ERR_ARG, "tail_counts is NULL");
    GeneralCfg cfg; cfg.ks = c->ks; cfg.S = c->S; cfg.classify = true; cfg.keep_all = true;
    cfg.argmax_n = argmax_refs < c->ix.nref ? argmax_refs : 0;
    apply_depth_cfg(c, cfg);
    GeneralOut go; go.out4 = out4; go.tail_counts = cfg.argmax_n ? tail_counts : nullptr;
    return general_run(c, bases, nullptr, offsets, nreads, cfg, go);
}

extern "C" int rk_classify_batch_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads,
                                        void* d_out4, uint32_t max_read_len, void* hip_stream) {
    if (!c || nreads < 0 || (nreads > 0 && (!d_bases || !d_offs || !d_out4))) return fail(RK_ERR_ARG, "bad arguments");
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    RKCHK(set_dev(c));
    if (nreads == 0) return RK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    // the resident-input entry point serves reads the fused kernel can take (len <= FUSED_MAXLEN and all
    // hashes inside the sketch); anything else is flagged -2 in d_out4 for the caller (rk_classify_batch
    // reroutes those through the general path itself).
    if (max_read_len == 0) RKCHK(device_max_len(c, d_offs, nreads, st, &max_read_len));
    return fused_device(c, d_bases, d_offs, nreads, d_out4, max_read_len, 0, nullptr, st);
}

// Same contract as rk_classify_batch_device, but no row is left flagged: rows the fused kernel hands back (long reads,
// reads with more windows than the sketch keeps, ...) are answered by the general kernels on the resident bases -- only
// the 4-byte offsets and the flagged rows cross PCIe.  Synchronises `hip_stream` (it has to look at the flags).
extern "C" int rk_classify_batch_device_all(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads,
                                            void* d_out4, uint32_t max_read_len, void* hip_stream) {
    RKCHK(rk_classify_batch_device(c, d_bases, d_offs, nreads, d_out4, max_read_len, hip_stream));
    if (nreads == 0) return RK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    std::vector<int32_t> rows((size_t)nreads * 4);
    std::vector<uint32_t> offs32((size_t)nreads + 1);
    HIPCHK(hipMemcpyAsync(rows.data(), d_out4, (size_t)nreads * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(offs32.data(), d_offs, ((size_t)nreads + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> idx;
    for (int64_t i = 0; i < nreads; ++i) if (rows[(size_t)i * 4] == -2) idx.push_back((uint32_t)i);
    if (idx.empty()) return RK_OK;
    const size_t m = idx.size();
    std::vector<uint64_t> lens_ps(m + 1, 0), starts(m);
    for (size_t j = 0; j < m; ++j) {
        starts[j] = offs32[idx[j]];
        lens_ps[j + 1] = lens_ps[j] + (uint64_t)(offs32[(size_t)idx[j] + 1] - offs32[idx[j]]);
    }
    std::vector<int32_t> res(m * 4);
    GeneralCfg cfg; cfg.ks = c->ks; cfg.S = c->S; cfg.classify = true; cfg.abs_starts = starts.data();
    apply_depth_cfg(c, cfg);
    GeneralOut go; go.out4 = res.data();
    RKCHK(general_run(c, nullptr, (const uint8_t*)d_bases, lens_ps.data(), (int64_t)m, cfg, go));
    // scatter the answers into the caller's result buffer
    RKCHK(c->w_ids.reserve(m * 4));
    RKCHK(c->w_out.reserve(m * 16));
    HIPCHK(hipMemcpyAsync(c->w_ids.p, idx.data(), m * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->w_out.p, res.data(), m * 16, hipMemcpyHostToDevice, st));
    HIPCHK(launch_scatter_rows(c->w_out.as<int32_t>(), c->w_ids.as<uint32_t>(), (uint32_t)m, (int32_t*)d_out4, st));
    HIPCHK(hipStreamSynchronize(st));
    return RK_OK;
}

extern "C" int rk_count_batch_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads,
                                     rk_counter* counter, void* hip_stream) {
    if (!c || !counter || nreads < 0 || (nreads > 0 && (!d_bases || !d_offs))) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(c));
    if (nreads == 0) return RK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    uint32_t ml = 0, end_off = 0;
    RKCHK(device_max_len(c, d_offs, nreads, st, &ml, &end_off));
    if (ml > (uint32_t)FUSED_MAXLEN) return fail(RK_ERR_LIMIT, "rk_count_batch_device: reads longer than %d need rk_count_batch", FUSED_MAXLEN);
    if (c->ks.n == 0) return fail(RK_ERR_STATE, "k-mer sizes unknown: call rk_set_references first");
    return fused_device(c, d_bases, d_offs, nreads, nullptr, ml, 1, counter, st, end_off);
}

// double-buffered host pipeline around the fused kernel. mode 0 classify, mode 1 count.
// Page-locked inputs (rk_host_alloc / hipHostMalloc: what the FASTQ front end fills) are read by the DMA engine where they lie;
// pageable ones go through the context's pinned staging buffers, copied by host_threads() threads while the previous chunk is on
// the link.  Results land directly in out4 when that is page-locked.  *flagged receives the number of rows the kernel handed back.
static int host_pipeline(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int32_t* out4,
                         int mode, rk_counter* count_into, int64_t* flagged = nullptr) {
    RKCHK(set_dev(c));
    // reads per chunk: the last chunk's kernel and D2H are not overlapped with anything, and a chunk's H2D cannot start before the
    // chunk two places earlier has left its slot, so shorter chunks finish sooner.  Measured (150 bp reads from page-locked buffers;
    // chunks of 2 M / 512 k / 128 k reads): 4 M reads 278 / 292 / 252 M reads/s, 16 M reads 300 / 313 M reads/s.  RKMH_CHUNK_READS overrides.
    // (The count pass keeps chunks of 2 M reads: its slot-partitioned form streams the whole table once per launch, rk_count.hip.)
    static const int64_t CHUNK_ENV = [] { const char* e = getenv("RKMH_CHUNK_READS"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 4096 ? v : 0); }();
    const int64_t MAX_READS = CHUNK_ENV ? CHUNK_ENV : (mode == 1 ? (int64_t)1 << 21 : (int64_t)1 << 19);
    const uint64_t MAX_BASES = 1ull << 29;
    bool src_pinned = nreads > 0 && caller_pinned_host(bases + offsets[0], (size_t)(offsets[nreads] - offsets[0]) + 4);
    bool out_pinned = mode == 0 && nreads > 0 && caller_pinned_host(out4, (size_t)nreads * 16);
    // pageable buffers of some size are page-locked for this call instead of being copied through the staging buffers
    ScopedHostRegister reg_src(nreads > 0 && !src_pinned ? bases + offsets[0] : nullptr, nreads > 0 ? (size_t)(offsets[nreads] - offsets[0]) + 4 : 0, (size_t)8 << 20);
    ScopedHostRegister reg_out(mode == 0 && nreads > 0 && !out_pinned ? out4 : nullptr, (size_t)nreads * 16, (size_t)4 << 20);
    src_pinned = src_pinned || reg_src.ok;
    out_pinned = out_pinned || reg_out.ok;
    int64_t i0 = 0, nflag = 0;
    int which = 0;
    auto drain = [&](Slot& s) -> int {
        if (!s.busy) return RK_OK;
        HIPCHK(hipEventSynchronize(s.done));
        if (mode == 0) {
            int32_t* dst = out4 + s.first * 4;
            const int32_t* src = out_pinned ? dst : s.h_out.as<int32_t>();
            std::vector<int64_t> part((size_t)host_threads() + 1, 0);
            std::atomic<int> slot_no{0};
            par_for((size_t)s.n, (size_t)1 << 17, [&](size_t lo, size_t hi) { // copy out (unless the DMA wrote in place) and count the rows handed back
                if (!out_pinned) memcpy(dst + lo * 4, src + lo * 4, (hi - lo) * 16);
                int64_t k = 0;
                for (size_t i = lo; i < hi; ++i) k += src[i * 4] == -2;
                part[(size_t)slot_no.fetch_add(1) % part.size()] += k;
            });
            for (int64_t k : part) nflag += k;
        }
        s.busy = false;
        return RK_OK;
    };
    while (i0 < nreads) {
        // a chunk: at most MAX_READS reads / MAX_BASES bases (offsets are monotone: the end is found by bisection, the longest read
        // by a parallel scan)
        int64_t i1 = std::min(nreads, i0 + MAX_READS);
        const uint64_t b0 = offsets[i0];
        if (offsets[i1] - b0 > MAX_BASES) {
            int64_t lo = i0 + 1, hi = i1;
            while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (offsets[mid] - b0 <= MAX_BASES) lo = mid; else hi = mid - 1; }
            i1 = lo;
        }
        const int64_t cn = i1 - i0;
        const uint64_t cb = offsets[i1] - b0;
        if (cb > 0xfffffff0ull) return fail(RK_ERR_LIMIT, "read %lld too long for a 32-bit batch", (long long)i0);
        Slot& s = c->slot[which];
        RKCHK(drain(s));
        RKCHK(s.h_offs.reserve((size_t)(cn + 1) * 4));
        if (!src_pinned) RKCHK(s.h_bases.reserve(cb + 64));
        if (!out_pinned && mode == 0) RKCHK(s.h_out.reserve((size_t)cn * 16));
        RKCHK(s.d_bases.reserve(cb + 64)); RKCHK(s.d_offs.reserve((size_t)(cn + 1) * 4)); RKCHK(s.d_out.reserve((size_t)cn * 16));
        uint32_t* ho = s.h_offs.as<uint32_t>();
        std::atomic<uint32_t> maxlen_a{0};
        par_for((size_t)cn + 1, (size_t)1 << 17, [&](size_t lo, size_t hi) { // 32-bit offsets relative to the chunk + the longest read
            uint32_t ml = 0;
            for (size_t i = lo; i < hi; ++i) {
                ho[i] = (uint32_t)(offsets[(size_t)i0 + i] - b0);
                if (i < (size_t)cn) { const uint64_t len = offsets[(size_t)i0 + i + 1] - offsets[(size_t)i0 + i]; if (len > ml) ml = (uint32_t)std::min<uint64_t>(len, 0xffffffffull); }
            }
            uint32_t cur = maxlen_a.load();
            while (ml > cur && !maxlen_a.compare_exchange_weak(cur, ml)) {}
        });
        const uint32_t maxlen = maxlen_a.load();
        const void* hsrc = bases + b0;
        if (!src_pinned) { par_memcpy(s.h_bases.p, bases + b0, cb); hsrc = s.h_bases.p; }
        HIPCHK(hipMemcpyAsync(s.d_bases.p, hsrc, cb, hipMemcpyHostToDevice, s.st));
        HIPCHK(hipMemcpyAsync(s.d_offs.p, s.h_offs.p, (size_t)(cn + 1) * 4, hipMemcpyHostToDevice, s.st));
        RKCHK(fused_device(c, s.d_bases.p, s.d_offs.p, cn, s.d_out.p, maxlen, mode, count_into, s.st, cb));
        if (mode == 0) HIPCHK(hipMemcpyAsync(out_pinned ? (void*)(out4 + i0 * 4) : s.h_out.p, s.d_out.p, (size_t)cn * 16, hipMemcpyDeviceToHost, s.st));
        HIPCHK(hipEventRecord(s.done, s.st));
        s.first = i0; s.n = cn; s.busy = true;
        which ^= 1;
        i0 = i1;
    }
    RKCHK(drain(c->slot[0]));
    RKCHK(drain(c->slot[1]));
    if (flagged) *flagged = nflag;
    return RK_OK;
}

// page-locked host memory for callers that want rk_classify_batch / rk_count_batch to run at link speed (no staging copy)
extern "C" int rk_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(RK_ERR_ARG, "out is NULL");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(RK_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
    return RK_OK;
}
// page-lock caller memory that is only read (a mapping of an input file): the DMA engines then read it in place
extern "C" int rk_host_register_readonly(const void* p, size_t bytes) {
    if (!p || !bytes) return fail(RK_ERR_ARG, "bad arguments");
    hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterReadOnly);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault); }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(RK_ERR_HIP, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    return RK_OK;
}
extern "C" void rk_host_unregister(const void* p) { if (p) { hipError_t e = hipHostUnregister(const_cast<void*>(p)); (void)e; } }
extern "C" void rk_host_free(void* p) { if (p) { hipError_t e = hipHostFree(p); (void)e; } }

extern "C" int rk_classify_batch(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int32_t* out4) {
    if (!c || !offsets || nreads < 0 || (nreads > 0 && !out4)) return fail(RK_ERR_ARG, "bad arguments");
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    if (nreads == 0) return RK_OK;
    // Reads the fused kernel is certain to hand back (longer than it stages, or with more windows than the sketch keeps,
    // so that bottom-S selection matters) go to the general path directly instead of being uploaded and hashed twice.
    // This only routes: the fused kernel still flags whatever it cannot answer exactly.
    auto general_only = [&](int64_t i) {
        const uint64_t len = offsets[i + 1] - offsets[i];
        if (len > (uint64_t)FUSED_MAXLEN) return true;
        uint64_t nw = 0;
        for (int j = 0; j < c->ks.n; ++j) nw += (uint64_t)num_windows((int)len, c->ks.k[j], c->pol.drop_last_window);
        return nw > (uint64_t)c->S;
    };
    std::atomic<int64_t> ngen_a{0};
    par_for((size_t)nreads, (size_t)1 << 17, [&](size_t lo, size_t hi) {
        int64_t k = 0;
        for (size_t i = lo; i < hi; ++i) k += general_only((int64_t)i) ? 1 : 0;
        ngen_a += k;
    });
    const int64_t ngen = ngen_a.load();
    if (ngen == nreads || !classify_tile_supported(c->ix.nref, 1)) { // e.g. a nanopore batch: one pass through the general path
        GeneralCfg cfg; cfg.ks = c->ks; cfg.S = c->S; cfg.classify = true;
        apply_depth_cfg(c, cfg);
        GeneralOut go; go.out4 = out4;
        return general_run(c, bases, nullptr, offsets, nreads, cfg, go);
    }
    if (ngen * 8 > nreads) { // mixed batch: the short reads are gathered for the fused kernel, the rest marked for the general path
        std::vector<int64_t> idx;
        idx.reserve((size_t)(nreads - ngen));
        for (int64_t i = 0; i < nreads; ++i) {
            if (general_only(i)) out4[i * 4] = -2;
            else idx.push_back(i);
        }
        std::vector<uint64_t> offs(idx.size() + 1, 0);
        for (size_t j = 0; j < idx.size(); ++j) offs[j + 1] = offs[j] + (offsets[idx[j] + 1] - offsets[idx[j]]);
        std::vector<uint8_t> sub((size_t)offs.back() + 64);
        for (size_t j = 0; j < idx.size(); ++j) memcpy(sub.data() + offs[j], bases + offsets[idx[j]], (size_t)(offs[j + 1] - offs[j]));
        std::vector<int32_t> res(idx.size() * 4);
        RKCHK(host_pipeline(c, sub.data(), offs.data(), (int64_t)idx.size(), res.data(), 0, nullptr));
        for (size_t j = 0; j < idx.size(); ++j) memcpy(out4 + idx[j] * 4, res.data() + j * 4, 16);
        return reroute_flagged(c, bases, offsets, nreads, out4);
    }
    int64_t nflag = 0;
    RKCHK(host_pipeline(c, bases, offsets, nreads, out4, 0, nullptr, &nflag));
    return nflag ? reroute_flagged(c, bases, offsets, nreads, out4) : RK_OK; // the pipeline counted the rows the kernel handed back
}

extern "C" int rk_count_batch(rk_ctx* c, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, rk_counter* counter) {
    if (!c || !offsets || nreads < 0 || !counter) return fail(RK_ERR_ARG, "bad arguments");
    if (c->ks.n == 0) return fail(RK_ERR_STATE, "k-mer sizes unknown: call rk_set_references first");
    if (nreads == 0) return RK_OK;
    // reads longer than the fused kernel's limit go through the tile hasher
    bool any_long = false;
    for (int64_t i = 0; i < nreads; ++i) if (offsets[i + 1] - offsets[i] > (uint64_t)FUSED_MAXLEN) { any_long = true; break; }
    if (!any_long) return host_pipeline(c, bases, offsets, nreads, nullptr, 1, counter);
    if (counter->compact) return fail(RK_ERR_NEED_FULL, "reads longer than %d bases: a compact depth map only counts reads that fit the sketch", FUSED_MAXLEN);
    RKCHK(counter_settle(counter));
    GeneralCfg cfg; cfg.ks = c->ks; cfg.inc_counter = counter;
    GeneralOut none;
    return general_run(c, bases, nullptr, offsets, nreads, cfg, none);
}

static inline char* put_int(char* w, int v) {
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *w++ = '-';
    while (n) *w++ = tmp[--n];
    return w;
}

extern "C" int rk_format_stream_line(char* dst, size_t cap, const char* ref_name, const char* read_name,
                                     int max_shared, int diff, int min_num, int sketch_size, int min_matches, int min_diff) {
    // src/rkmh.cpp:887-892: ref \t read \t max_shared \t sketch_size<depth filter> \t <match filter> \t <diff filter> \n
    const bool diff_filter = diff > min_diff;
    const bool depth_filter = min_num <= min_matches;
    const bool match_filter = max_shared < min_matches;
    const size_t ln = strlen(ref_name), lq = strlen(read_name);
    if (ln + lq + 64 >= cap) return fail(RK_ERR_ARG, "line buffer too small");
    char* w = dst;
    memcpy(w, ref_name, ln); w += ln; *w++ = '\t';
    memcpy(w, read_name, lq); w += lq; *w++ = '\t';
    w = put_int(w, max_shared); *w++ = '\t';
    w = put_int(w, sketch_size);
    if (depth_filter) { memcpy(w, "FAIL:DEPTH", 10); w += 10; }
    *w++ = '\t';
    if (match_filter) { memcpy(w, "FAIL:MATCHES", 12); w += 12; }
    *w++ = '\t';
    if (!diff_filter) { memcpy(w, "FAIL:DIFF", 9); w += 9; }
    *w++ = '\n';
    *w = '\0';
    return (int)(w - dst);
}

// ---- call ------------------------------------------------------------------------------------------
extern "C" int rk_call(rk_ctx* c, const uint8_t* ref_bases, const uint64_t* ref_offsets, int nref,
                       const uint8_t* read_bases, const uint64_t* read_offsets, int64_t nreads, int k, int window_len,
                       rk_call_record** out, int64_t* nout) {
    static_assert(sizeof(rk_call_record) == sizeof(CallRecord), "record layouts must match");
    if (!c || !ref_offsets || !read_offsets || nref < 1 || nreads < 0 || !out || !nout) return fail(RK_ERR_ARG, "bad arguments");
    if (k < 1 || k > RK_MAX_K) return fail(RK_ERR_LIMIT, "k=%d outside [1,%d]", k, RK_MAX_K);
    if (window_len < 1) return fail(RK_ERR_ARG, "window length must be positive");
    RKCHK(set_dev(c));
    *out = nullptr; *nout = 0;
    GeneralCfg cfg; cfg.ks.n = 1; cfg.ks.k[0] = k;
    // windows of the reads / of the references
    uint64_t wr = 0;
    for (int64_t i = 0; i < nreads; ++i) wr += (uint64_t)num_windows((int)(read_offsets[i + 1] - read_offsets[i]), k, c->pol.drop_last_window);
    std::vector<uint64_t> win_off((size_t)nref + 1, 0);
    for (int i = 0; i < nref; ++i)
        win_off[(size_t)i + 1] = win_off[(size_t)i] + (uint64_t)num_windows((int)(ref_offsets[i + 1] - ref_offsets[i]), k, c->pol.drop_last_window);
    const uint64_t wtot = win_off[(size_t)nref];
    if (wtot >= (1ull << 30)) return fail(RK_ERR_LIMIT, "more than 2^30 reference positions");
    // exact depth map
    uint64_t cap = 1024;
    while (cap < 2 * wr) cap <<= 1;
    DevBuf d_keys, d_counts, d_depth, d_prefix, d_scratch, d_ref, d_refoff, d_winoff, d_rec, d_cnt;
    int rc = RK_OK;
    auto cleanup = [&]() { for (DevBuf* b : {&d_keys, &d_counts, &d_depth, &d_prefix, &d_scratch, &d_ref, &d_refoff, &d_winoff, &d_rec, &d_cnt}) b->release(); };
#define CALLCHK(expr) do { rc = (expr); if (rc != RK_OK) { cleanup(); return rc; } } while (0)
#define CALLHIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } } while (0)
    CALLCHK(d_keys.reserve(cap * 8));
    CALLCHK(d_counts.reserve(cap * 4 + 16));
    CALLHIP(hipMemsetAsync(d_keys.p, 0, cap * 8, c->st));
    CALLHIP(hipMemsetAsync(d_counts.p, 0, cap * 4 + 16, c->st));
    DepthTable t;
    t.keys = d_keys.as<uint64_t>(); t.counts = d_counts.as<uint32_t>(); t.zero_count = d_counts.as<uint32_t>() + cap; t.mask = cap - 1;
    {   // pass over the reads: read_hash_to_depth[h] += 1 (rkmh.cpp:1613-1622)
        GeneralCfg c1 = cfg; c1.depth_insert = &t;
        GeneralOut none;
        CALLCHK(general_run(c, read_bases, nullptr, read_offsets, nreads, c1, none));
    }
    // references: upper-cased copy stays on the device for the candidate enumeration
    const uint64_t rbytes = ref_offsets[nref];
    CALLCHK(d_ref.reserve(rbytes + 64));
    CALLCHK(d_refoff.reserve(((size_t)nref + 1) * 8));
    CALLCHK(d_winoff.reserve(((size_t)nref + 1) * 8));
    CALLCHK(d_depth.reserve((wtot + 1) * 4));
    CALLCHK(d_prefix.reserve((wtot + 2) * 8));
    CALLCHK(d_scratch.reserve(((wtot / 1024 + 2) * 2 + 4200) * 8));
    if (rbytes) CALLHIP(hipMemcpyAsync(d_ref.p, ref_bases, rbytes, hipMemcpyHostToDevice, c->st));
    CALLHIP(launch_to_upper(d_ref.as<uint8_t>(), rbytes, c->st));
    CALLHIP(hipMemcpyAsync(d_refoff.p, ref_offsets, ((size_t)nref + 1) * 8, hipMemcpyHostToDevice, c->st));
    CALLHIP(hipMemcpyAsync(d_winoff.p, win_off.data(), ((size_t)nref + 1) * 8, hipMemcpyHostToDevice, c->st));
    {   // depth of every reference window, in reference order (rkmh.cpp:1785)
        GeneralCfg c2 = cfg; c2.depth_lookup = &t; c2.depth_out = d_depth.as<int32_t>();
        GeneralOut none;
        CALLCHK(general_run(c, ref_bases, nullptr, ref_offsets, nref, c2, none));
    }
    CALLHIP(hipMemsetAsync(d_prefix.p, 0, (wtot + 2) * 8, c->st));
    CALLHIP(launch_exclusive_scan(d_depth.as<int32_t>(), wtot, d_prefix.as<int64_t>(), d_scratch.as<int64_t>(), c->st));
    uint32_t rcap = 1u << 16;
    CALLCHK(d_cnt.reserve(16));
    std::vector<rk_call_record> recs;
    for (;;) {
        CALLCHK(d_rec.reserve((size_t)rcap * sizeof(CallRecord)));
        CALLHIP(hipMemsetAsync(d_cnt.p, 0, 16, c->st));
        CALLHIP(launch_call_enumerate(d_ref.as<uint8_t>(), d_refoff.as<uint64_t>(), d_winoff.as<uint64_t>(), nref, wtot,
                                      d_depth.as<int32_t>(), d_prefix.as<int64_t>(), k, window_len, t, c->pol, d_rec.as<CallRecord>(),
                                      d_cnt.as<uint32_t>(), rcap, c->st));
        uint32_t n = 0;
        CALLHIP(hipMemcpyAsync(&n, d_cnt.p, 4, hipMemcpyDeviceToHost, c->st));
        CALLHIP(hipStreamSynchronize(c->st));
        if (n > rcap) { rcap = n + 1024; continue; } // rare: more calls than expected, run again with room for all
        recs.resize(n);
        if (n) CALLHIP(hipMemcpy(recs.data(), d_rec.p, (size_t)n * sizeof(CallRecord), hipMemcpyDeviceToHost));
        break;
    }
    cleanup();
#undef CALLCHK
#undef CALLHIP
    rk_call_record* r = (rk_call_record*)malloc(sizeof(rk_call_record) * (recs.empty() ? 1 : recs.size()));
    if (!r) return fail(RK_ERR_NOMEM, "malloc");
    if (!recs.empty()) memcpy(r, recs.data(), recs.size() * sizeof(rk_call_record));
    *out = r; *nout = (int64_t)recs.size();
    return RK_OK;
}


// ------------------------------------------------------------------------------------------------
// FASTQ text parsed on the device (rk_fastq.hip): one slot = one block in flight (its own stream, page-locked text buffer, device
// arrays).  Several slots of one context may be driven from several host threads at once.
struct rk_fastq_slot {
    rk_ctx* c = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr; // blocking-sync event: a dozen host threads wait for their blocks at once and must SLEEP while they
                             // do (hipStreamSynchronize spins: the waiting threads would take the cores the reading and formatting ones need)
    uint64_t max_bytes = 0;
    PinBuf h_text, h_out4, h_spans, h_info;
    DevBuf d_text, d_u32, d_bases, d_out4, d_scan;
    FqDev d{};
    uint64_t pending = 0;   // bytes of the block between submit and finish
    bool submitted = false;
    // BGZF members inflated on the device (rk_fastq_slot_load_bgzf): compressed bytes + member table up, text built in d_inf, the
    // job's records moved to d_text -- the next submit / count then skips its upload (text_on_device)
    PinBuf h_comp, h_mem;
    DevBuf d_comp, d_mem, d_inf, d_match;
    bool text_on_device = false;
    struct { bool pending = false; int64_t b0 = 0, b1 = 0, nb = 0; uint64_t u_lo = 0, ntext = 0; uint32_t nm = 0; } inf; // between load_bgzf_begin and _end
    // rk_fastq_slot_set_source: the block's text lies in caller memory (a page-locked mapping of the file): the next submit uploads it
    // from there, and the slot reads the text there where it needs it on the host (rerouted reads)
    const uint8_t* src = nullptr;      // of the block in flight (nullptr: h_text)
    const uint8_t* next_src = nullptr; // armed for the next submit
};

extern "C" void rk_fastq_slot_destroy(rk_fastq_slot* s) {
    if (!s) return;
    if (s->c) { hipError_t e = hipSetDevice(s->c->device); (void)e; }
    if (s->st) { hipError_t e = hipStreamSynchronize(s->st); (void)e; e = hipStreamDestroy(s->st); (void)e; }
    if (s->ev) { hipError_t e = hipEventDestroy(s->ev); (void)e; }
    for (PinBuf* b : {&s->h_text, &s->h_out4, &s->h_spans, &s->h_info, &s->h_comp, &s->h_mem}) b->release();
    for (DevBuf* b : {&s->d_text, &s->d_u32, &s->d_bases, &s->d_out4, &s->d_scan, &s->d_comp, &s->d_mem, &s->d_inf, &s->d_match}) b->release();
    delete s;
}

extern "C" int rk_fastq_slot_create(rk_ctx* c, uint64_t max_bytes, rk_fastq_slot** out) {
    if (!c || !out || max_bytes < 4096 || max_bytes > ((uint64_t)1 << 31)) return fail(RK_ERR_ARG, "bad arguments (block size 4 KB .. 2 GB)");
    RKCHK(set_dev(c));
    rk_fastq_slot* s = new rk_fastq_slot();
    s->c = c; s->max_bytes = max_bytes;
    struct Guard { rk_fastq_slot* s; ~Guard() { if (s) rk_fastq_slot_destroy(s); } } guard{s};
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->ev, hipEventBlockingSync | hipEventDisableTiming));
    // capacities: records of fewer than 64 bytes on average (reads of about 25 bases) make the block "irregular" (FQ_BAD_CAP) --
    // the host scanner takes it -- instead of sizing every array, page-locked ones included, for the worst case
    const uint32_t chunks = (uint32_t)((max_bytes + 4095) / 4096);
    const uint32_t rec_cap = (uint32_t)(max_bytes / 64 + 64), line_cap = 4 * rec_cap + 16;
    RKCHK(s->h_text.reserve(max_bytes + 64));
    RKCHK(s->h_out4.reserve((size_t)rec_cap * 16));
    RKCHK(s->h_spans.reserve((size_t)rec_cap * 20));
    RKCHK(s->h_info.reserve(16));
    RKCHK(s->d_text.reserve(max_bytes + 64));
    RKCHK(s->d_bases.reserve(max_bytes + 64));
    RKCHK(s->d_out4.reserve((size_t)rec_cap * 16));
    const size_t n32 = (size_t)2 * (chunks + 1) + line_cap + (size_t)6 * (rec_cap + 1) + 4;
    RKCHK(s->d_u32.reserve(n32 * 4));
    const size_t tb = fq_scan_temp_bytes(std::max(chunks + 1, rec_cap + 1));
    RKCHK(s->d_scan.reserve(tb));
    uint32_t* u = s->d_u32.as<uint32_t>();
    FqDev& d = s->d;
    d.chunk_cnt = u; u += chunks + 1;
    d.chunk_base = u; u += chunks + 1;
    d.nl = u; u += line_cap;
    d.seq_off = u; u += rec_cap + 1;
    d.seq_len = u; u += rec_cap + 1;
    d.qual_off = u; u += rec_cap + 1;
    d.name_off = u; u += rec_cap + 1;
    d.name_len = u; u += rec_cap + 1;
    d.out_off = u; u += rec_cap + 1;
    d.info = u;
    d.line_cap = line_cap; d.rec_cap = rec_cap;
    d.bases = s->d_bases.as<uint8_t>();
    d.scan_tmp = s->d_scan.p; d.scan_tmp_bytes = tb;
    HIPCHK(hipMemsetAsync(s->d_u32.p, 0, n32 * 4, s->st)); // stale lengths past a block's last record must at least be defined
    HIPCHK(hipStreamSynchronize(s->st));
    guard.s = nullptr;
    *out = s;
    return RK_OK;
}

extern "C" uint8_t* rk_fastq_slot_text(rk_fastq_slot* s) { return s ? s->h_text.as<uint8_t>() : nullptr; }
// The NEXT block of this slot is read from `text` (caller memory that stays valid and unchanged until the block's finish / count
// has returned) instead of the slot's own buffer: a page-locked mapping of the input file (mmap + hipHostRegister) lets the DMA
// engine read the page cache itself -- no pread copy (tools/ubench/mmap_register.hip: 55 GB/s against 18-20 for one thread's pread + upload).
extern "C" int rk_fastq_slot_set_source(rk_fastq_slot* s, const uint8_t* text) {
    if (!s) return fail(RK_ERR_ARG, "slot is NULL");
    s->next_src = text;
    return RK_OK;
}

// A BGZF job inflated ON THE DEVICE (rk_inflate.hip): the compressed bytes of members [b0 - 1, b1 + 2) go up -- 0.58 x the text for
// level-1 FASTQ --, one wave per member inflates them, the first record starts at or after the text of b0 and of b1 are found by
// the four-line rule (k_fastq_first_start: the rule of rk_bgzf_fastq_records, so host-inflated and device-inflated jobs agree),
// and the records between them are moved to the slot's text buffer; a copy travels back to rk_fastq_slot_text() for the output
// formatters.  The NEXT rk_fastq_slot_submit / _classify / _count of this slot takes *nbytes and skips its upload.
// Returns RK_OK, or 1: this job is for the host route (rk_bgzf_fastq_records) -- a member the device could not inflate, text that
// does not begin with '@', a record that outgrows the lookahead or the slot.
extern "C" int rk_fastq_slot_load_bgzf_begin(rk_fastq_slot* s, const rk_bgzf* z, int64_t b0, int64_t b1) {
    static const bool timing = getenv("RKMH_BGZF_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    if (!s || !z || b0 < 0 || b1 <= b0 || b1 > rk_bgzf_members(z)) return fail(RK_ERR_ARG, "bad arguments");
    s->text_on_device = false;
    s->inf.pending = false;
    rk_ctx* c = s->c;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    const int64_t nb = rk_bgzf_members(z);
    const int64_t lo = b0 > 0 ? b0 - 1 : 0, ext = std::min<int64_t>(nb, b1 + 2);
    const uint32_t nm = (uint32_t)(ext - lo);
    uint64_t f_lo = 0, f_hi = 0;
    uint32_t tot = 0, hd = 0, us = 0;
    RKCHK(rk_bgzf_member(z, lo, &f_lo, &tot, &hd, &us));
    RKCHK(rk_bgzf_member(z, ext - 1, &f_hi, &tot, &hd, &us));
    const uint64_t cbytes = f_hi + tot - f_lo;
    const uint64_t u_lo = rk_bgzf_text_offset(z, lo), u_b0 = rk_bgzf_text_offset(z, b0), u_b1 = rk_bgzf_text_offset(z, b1), u_ext = rk_bgzf_text_offset(z, ext);
    const uint64_t ntext = u_ext - u_lo;
    if (ntext > s->max_bytes + 4 * 65536ull || cbytes >= ((uint64_t)1 << 31)) return 1;
    // (sized for the slot, not for this job: growing a page-locked buffer by a few kilobytes per job costs ~100 ms each time, and
    // every job of a file is a little different -- 5/8 of the text covers level-1 FASTQ, a member is at most 64 KB of text)
    const uint64_t cap_text = s->max_bytes + 5 * 65536ull + 64, cap_mem = std::max<uint64_t>(nm, cap_text / 32768 + 16);
    RKCHK(s->h_comp.reserve(std::max<uint64_t>(cbytes + 160, cap_text * 5 / 8)));
    RKCHK(s->h_mem.reserve((size_t)cap_mem * sizeof(InflateMember) + (size_t)cap_mem * 8 + 64));
    RKCHK(s->d_comp.reserve(std::max<uint64_t>(cbytes + 160, cap_text * 5 / 8)));
    RKCHK(s->d_mem.reserve((size_t)cap_mem * sizeof(InflateMember) + (size_t)cap_mem * 8 + 64));
    RKCHK(s->d_inf.reserve(cap_text));
    const double t_reserve = ms_since(t_0);
    memcpy(s->h_comp.p, rk_bgzf_image(z) + f_lo, cbytes);
    const double t_copy = ms_since(t_0);
    memset(s->h_comp.as<uint8_t>() + cbytes, 0, 80);
    InflateMember* mt = s->h_mem.as<InflateMember>();
    uint64_t scratch_dw = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        uint64_t fo = 0;
        RKCHK(rk_bgzf_member(z, lo + i, &fo, &tot, &hd, &us));
        mt[i].in_off = (uint32_t)(fo - f_lo) + hd; mt[i].in_len = tot - hd - 8;
        mt[i].out_off = (uint32_t)(rk_bgzf_text_offset(z, lo + i) - u_lo); mt[i].out_len = us;
        mt[i].match_off = (uint32_t)scratch_dw; mt[i].pad = 0;
        scratch_dw += inflate_scratch_dwords(us);
    }
    RKCHK(s->d_match.reserve(std::max<uint64_t>(scratch_dw * 4 + 64, cap_text * 5 / 2 + cap_mem * 32)));
    const size_t cpad = ((cbytes + 15) & ~(size_t)15) + 64; // (the lanes of pass 1 request whole 16-byte pieces a little past their member)
    HIPCHK(hipMemcpyAsync(s->d_comp.p, s->h_comp.p, cpad, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->d_mem.p, mt, (size_t)nm * sizeof(InflateMember), hipMemcpyHostToDevice, st));
    uint32_t* d_status = reinterpret_cast<uint32_t*>(s->d_mem.as<uint8_t>() + (((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15));
    uint32_t* h_status = reinterpret_cast<uint32_t*>(s->h_mem.as<uint8_t>() + (((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15));
    HIPCHK(launch_inflate_members(s->d_comp.as<uint8_t>(), (uint32_t)cpad, s->d_mem.as<InflateMember>(), nm, s->d_inf.as<uint8_t>(), s->d_match.as<uint32_t>(), d_status, st));
    // the cuts: cuts[0] = head, cuts[1] = tail (in the inflated text of members lo .. ext)
    uint32_t* d_cuts = s->d.info; // (the index kernels write it afterwards)
    const bool at_eof = ext == nb;
    if (b0 > 0) HIPCHK(launch_fastq_first_start(s->d_inf.as<uint8_t>(), (uint32_t)ntext, (uint32_t)(u_b0 - u_lo), 1u << 18, at_eof, d_cuts, 0, st));
    if (b1 < nb) HIPCHK(launch_fastq_first_start(s->d_inf.as<uint8_t>(), (uint32_t)ntext, (uint32_t)(u_b1 - u_lo), 1u << 18, at_eof, d_cuts, 1, st));
    uint32_t* h_info = s->h_info.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(h_info, d_cuts, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_status, d_status, (size_t)nm * 4, hipMemcpyDeviceToHost, st));
    // (the first and the last byte of the text decide two small things on the host)
    HIPCHK(hipMemcpyAsync(h_info + 2, s->d_inf.as<uint8_t>(), 1, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(h_info + 2) + 1, s->d_inf.as<uint8_t>() + (ntext ? ntext - 1 : 0), 1, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s->ev, st));
    if (timing) fprintf(stderr, "[bgzf device] %u members, %.1f MB in, %.1f MB text: reserve %.1f ms, copy %.1f, enqueue %.1f\n", nm, cbytes / 1e6, ntext / 1e6,
                        t_reserve, t_copy - t_reserve, ms_since(t_0) - t_copy);
    s->inf.pending = true; s->inf.b0 = b0; s->inf.b1 = b1; s->inf.nb = nb; s->inf.u_lo = u_lo; s->inf.ntext = ntext; s->inf.nm = nm;
    return RK_OK;
}

extern "C" int rk_fastq_slot_load_bgzf_end(rk_fastq_slot* s, uint64_t* nbytes, uint64_t* text_off) {
    if (!s || !nbytes) return fail(RK_ERR_ARG, "bad arguments");
    *nbytes = 0;
    if (!s->inf.pending) return fail(RK_ERR_STATE, "rk_fastq_slot_load_bgzf_end without a begun job");
    s->inf.pending = false;
    rk_ctx* c = s->c;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    const int64_t b0 = s->inf.b0, b1 = s->inf.b1, nb = s->inf.nb;
    const uint64_t u_lo = s->inf.u_lo, ntext = s->inf.ntext;
    const uint32_t nm = s->inf.nm;
    const uint32_t* h_status = reinterpret_cast<const uint32_t*>(s->h_mem.as<uint8_t>() + (((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15));
    const uint32_t* h_info = s->h_info.as<uint32_t>();
    static const bool timing = getenv("RKMH_BGZF_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventSynchronize(s->ev));
    if (timing) fprintf(stderr, "[bgzf device] waited %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count());
    for (uint32_t i = 0; i < nm; ++i) if (h_status[i] != 0) return 1;
    const uint8_t first_byte = reinterpret_cast<const uint8_t*>(h_info + 2)[0], last_byte = reinterpret_cast<const uint8_t*>(h_info + 2)[1];
    uint64_t head = b0 > 0 ? h_info[0] : 0, tail = b1 < nb ? h_info[1] : ntext;
    if (head == 0xFFFFFFFFull || tail == 0xFFFFFFFFull) return 1;
    if (head > tail) head = tail;
    if (b0 == 0 && tail > 0 && first_byte != '@') return 1;
    uint64_t n = tail - head;
    if (n + 1 > s->max_bytes) return 1;
    if (text_off) *text_off = u_lo + head;
    if (n == 0) return RK_OK;
    HIPCHK(hipMemcpyAsync(s->d_text.p, s->d_inf.as<uint8_t>() + head, n, hipMemcpyDeviceToDevice, st));
    if (b1 == nb && tail == ntext && last_byte != '\n') { HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + n, '\n', 1, st)); ++n; } // a last line without its newline
    HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + n, 'A', 16, st)); // the index kernels read whole 16-byte pieces
    HIPCHK(hipMemcpyAsync(s->h_text.p, s->d_text.p, n, hipMemcpyDeviceToHost, st)); // names, sequences and qualities for the formatters
    s->text_on_device = true;
    *nbytes = n;
    return RK_OK;
}

extern "C" int rk_fastq_slot_load_bgzf(rk_fastq_slot* s, const rk_bgzf* z, int64_t b0, int64_t b1, uint64_t* nbytes, uint64_t* text_off) {
    if (!nbytes) return fail(RK_ERR_ARG, "bad arguments");
    *nbytes = 0;
    if (text_off && z && b0 >= 0 && b0 < rk_bgzf_members(z)) *text_off = rk_bgzf_text_offset(z, b0);
    const int rc = rk_fastq_slot_load_bgzf_begin(s, z, b0, b1);
    if (rc != RK_OK) return rc;
    return rk_fastq_slot_load_bgzf_end(s, nbytes, text_off);
}

// The two halves of rk_fastq_slot_classify, for callers that keep two slots per thread: submit() enqueues the upload and the
// index / check / pack kernels and returns at once; finish() waits for them, launches the classification and collects the rows.
// Between the two the caller can read its next block into its other slot -- the link and the GPU work while the host reads.
extern "C" int rk_fastq_slot_submit(rk_fastq_slot* s, uint64_t nbytes) {
    if (!s || nbytes > s->max_bytes) return fail(RK_ERR_ARG, "bad arguments");
    rk_ctx* c = s->c;
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    s->pending = nbytes;
    s->submitted = true;
    if (nbytes == 0) return RK_OK;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    s->src = nullptr;
    if (s->text_on_device) { s->text_on_device = false; s->next_src = nullptr; } // rk_fastq_slot_load_bgzf left this block's text in d_text (and on its way to h_text)
    else if (s->next_src) { // straight from the caller's (page-locked) memory: no copy into the slot's buffer
        s->src = s->next_src; s->next_src = nullptr;
        HIPCHK(hipMemcpyAsync(s->d_text.p, s->src, nbytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + nbytes, 'A', 16, st)); // the device reads whole 16-byte pieces
    } else {
        uint8_t* text = s->h_text.as<uint8_t>();
        memset(text + nbytes, 'A', 16); // the device reads whole 16-byte pieces
        HIPCHK(hipMemcpyAsync(s->d_text.p, text, (nbytes + 15) & ~(uint64_t)15, hipMemcpyHostToDevice, st));
    }
    HIPCHK(launch_fastq_index(s->d, s->d_text.as<uint8_t>(), nbytes, st));
    HIPCHK(hipMemcpyAsync(s->h_info.p, s->d.info, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s->ev, st));
    return RK_OK;
}

extern "C" int rk_fastq_slot_finish(rk_fastq_slot* s, rk_fastq_result* res) {
    if (!s || !res) return fail(RK_ERR_ARG, "bad arguments");
    if (!s->submitted) return fail(RK_ERR_STATE, "rk_fastq_slot_finish without rk_fastq_slot_submit");
    s->submitted = false;
    rk_ctx* c = s->c;
    memset(res, 0, sizeof *res);
    if (s->pending == 0) return RK_OK;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    const uint8_t* text = s->src ? s->src : s->h_text.as<uint8_t>();
    uint32_t* info = s->h_info.as<uint32_t>();
    HIPCHK(hipEventSynchronize(s->ev));
    if (info[0] != 0) { res->status = (int32_t)info[0]; return RK_OK; } // not strictly four lines per record: the caller's scanner takes the block
    const int64_t nrec = (int64_t)info[1];
    res->nrec = nrec;
    if (nrec == 0) return RK_OK;
    int32_t* out4 = s->h_out4.as<int32_t>();
    uint32_t* spans = s->h_spans.as<uint32_t>();
    RKCHK(fused_device(c, s->d.bases, s->d.out_off, nrec, s->d_out4.p, info[2], 0, nullptr, st));
    HIPCHK(hipMemcpyAsync(out4, s->d_out4.p, (size_t)nrec * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spans, s->d.name_off, (size_t)nrec * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spans + nrec, s->d.name_len, (size_t)nrec * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spans + 2 * nrec, s->d.seq_off, (size_t)nrec * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spans + 3 * nrec, s->d.seq_len, (size_t)nrec * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spans + 4 * nrec, s->d.qual_off, (size_t)nrec * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s->ev, st));
    HIPCHK(hipEventSynchronize(s->ev));
    res->out4 = out4;
    res->name_off = spans; res->name_len = spans + nrec; res->seq_off = spans + 2 * nrec; res->seq_len = spans + 3 * nrec;
    res->qual_off = spans + 4 * nrec;
    // rows the fused kernel handed back (long reads, more windows than the sketch keeps, ...): the general path, from the text
    std::vector<int64_t> idx;
    for (int64_t i = 0; i < nrec; ++i) if (out4[i * 4] == -2) idx.push_back(i);
    if (!idx.empty()) {
        std::vector<uint64_t> offs(idx.size() + 1, 0);
        for (size_t j = 0; j < idx.size(); ++j) offs[j + 1] = offs[j] + res->seq_len[idx[j]];
        std::vector<uint8_t> sub((size_t)offs.back() + 16);
        for (size_t j = 0; j < idx.size(); ++j) memcpy(sub.data() + offs[j], text + res->seq_off[idx[j]], res->seq_len[idx[j]]);
        std::vector<int32_t> rows(idx.size() * 4);
        GeneralCfg cfg; cfg.ks = c->ks; cfg.S = c->S; cfg.classify = true;
        apply_depth_cfg(c, cfg);
        GeneralOut go; go.out4 = rows.data();
        {
            std::lock_guard<std::mutex> lock(c->general_mu);
            RKCHK(general_run(c, sub.data(), nullptr, offs.data(), (int64_t)idx.size(), cfg, go));
        }
        for (size_t j = 0; j < idx.size(); ++j) memcpy(out4 + idx[j] * 4, rows.data() + j * 4, 16);
    }
    return RK_OK;
}

// Pass 1 of -M on a block of raw FASTQ text (rkmh.cpp:904-910): split, check and pack on the device as rk_fastq_slot_classify does,
// then count every window's hash into `counter`.  *status != 0: the block is not four lines per record and NOTHING was counted.
extern "C" int rk_fastq_slot_count(rk_fastq_slot* s, uint64_t nbytes, rk_counter* counter, int32_t* status, int64_t* nrec_out) {
    if (!s || !counter || !status) return fail(RK_ERR_ARG, "bad arguments");
    if (counter->ctx != s->c) return fail(RK_ERR_ARG, "the counter belongs to another context");
    *status = 0;
    if (nrec_out) *nrec_out = 0;
    RKCHK(rk_fastq_slot_submit(s, nbytes));
    s->submitted = false;
    if (nbytes == 0) return RK_OK;
    rk_ctx* c = s->c;
    RKCHK(set_dev(c));
    uint32_t* info = s->h_info.as<uint32_t>();
    HIPCHK(hipEventSynchronize(s->ev));
    if (info[0] != 0) { *status = (int32_t)info[0]; return RK_OK; }
    const int64_t nrec = (int64_t)info[1];
    if (nrec_out) *nrec_out = nrec;
    if (nrec == 0) return RK_OK;
    if (info[2] > (uint32_t)FUSED_MAXLEN && counter->compact)
        return fail(RK_ERR_NEED_FULL, "reads longer than %d bases: a compact depth map only counts reads that fit the sketch", FUSED_MAXLEN);
    if (info[2] > (uint32_t)FUSED_MAXLEN) {
        // a read longer than the fused kernel's limit: the whole block through the tile hasher, from the text (as rk_count_batch does)
        uint32_t* spans = s->h_spans.as<uint32_t>();
        HIPCHK(hipMemcpyAsync(spans, s->d.seq_off, (size_t)nrec * 4, hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipMemcpyAsync(spans + nrec, s->d.seq_len, (size_t)nrec * 4, hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipEventRecord(s->ev, s->st));
        HIPCHK(hipEventSynchronize(s->ev));
        const uint8_t* text = s->src ? s->src : s->h_text.as<uint8_t>();
        std::vector<uint64_t> offs((size_t)nrec + 1, 0);
        for (int64_t i = 0; i < nrec; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + spans[nrec + i];
        std::vector<uint8_t> sub((size_t)offs.back() + 16);
        for (int64_t i = 0; i < nrec; ++i) memcpy(sub.data() + offs[(size_t)i], text + spans[i], spans[nrec + i]);
        std::lock_guard<std::mutex> lock(c->general_mu);
        RKCHK(counter_settle(counter));
        GeneralCfg cfg; cfg.ks = c->ks; cfg.inc_counter = counter;
        GeneralOut none;
        return general_run(c, sub.data(), nullptr, offs.data(), nrec, cfg, none);
    }
    RKCHK(fused_device(c, s->d.bases, s->d.out_off, nrec, nullptr, info[2], 1, counter, s->st));
    HIPCHK(hipEventRecord(s->ev, s->st));
    HIPCHK(hipEventSynchronize(s->ev));
    return RK_OK;
}

extern "C" int rk_fastq_slot_classify(rk_fastq_slot* s, uint64_t nbytes, rk_fastq_result* res) {
    if (!res) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(rk_fastq_slot_submit(s, nbytes));
    return rk_fastq_slot_finish(s, res);
}

// ---- reference FASTA text stripped on the device (rk_fasta.hip) -------------------------------------------------------------
struct rk_fasta_load {
    rk_ctx* c = nullptr;
    uint64_t cap = 0;
    DevBuf d_text, d_bases, d_u32, d_u64, d_rec, d_names, d_scan, d_info;
    PinBuf h_small;
    FaDev d{};
    std::vector<uint64_t> offsets, name_offsets;
    std::vector<char> names;
    int64_t nseq = 0;
    bool finished = false;
};

extern "C" void rk_fasta_load_destroy(rk_fasta_load* L) {
    if (!L) return;
    if (L->c) { hipError_t e = hipSetDevice(L->c->device); (void)e; e = hipStreamSynchronize(L->c->st); (void)e; }
    for (DevBuf* b : {&L->d_text, &L->d_bases, &L->d_u32, &L->d_u64, &L->d_rec, &L->d_names, &L->d_scan, &L->d_info}) b->release();
    L->h_small.release();
    delete L;
}

extern "C" int rk_fasta_load_create(rk_ctx* c, uint64_t text_bytes, rk_fasta_load** out) {
    if (!c || !out || text_bytes < 1 || text_bytes > ((uint64_t)1 << 37)) return fail(RK_ERR_ARG, "bad arguments (1 byte .. 128 GB of text)");
    RKCHK(set_dev(c));
    rk_fasta_load* L = new rk_fasta_load();
    L->c = c; L->cap = text_bytes;
    struct Guard { rk_fasta_load* L; ~Guard() { if (L) rk_fasta_load_destroy(L); } } guard{L};
    const uint64_t chunks = fa_chunks(text_bytes);
    RKCHK(L->d_text.reserve(chunks * 4096 + 64)); // the kernels read whole 4 KB chunks
    RKCHK(L->d_u32.reserve(2 * chunks * 4 + 64));
    RKCHK(L->d_u64.reserve(4 * (chunks + 1) * 8 + 64));
    RKCHK(L->d_info.reserve(64));
    RKCHK(L->h_small.reserve(64));
    guard.L = nullptr;
    *out = L;
    return RK_OK;
}

// the first nbytes of the slot's page-locked text buffer become text[text_offset ..); returns when the buffer may be refilled
extern "C" int rk_fasta_load_put(rk_fasta_load* L, rk_fastq_slot* via, uint64_t text_offset, uint64_t nbytes) {
    if (!L || !via || L->finished) return fail(RK_ERR_ARG, "bad arguments");
    if (via->c->device != L->c->device) return fail(RK_ERR_ARG, "the slot belongs to another device");
    if (nbytes > via->max_bytes || text_offset > L->cap || nbytes > L->cap - text_offset) return fail(RK_ERR_ARG, "block outside the text");
    if (nbytes == 0) return RK_OK;
    RKCHK(set_dev(L->c));
    HIPCHK(hipMemcpyAsync(L->d_text.as<uint8_t>() + text_offset, via->h_text.p, nbytes, hipMemcpyHostToDevice, via->st));
    HIPCHK(hipEventRecord(via->ev, via->st));
    HIPCHK(hipEventSynchronize(via->ev));
    return RK_OK;
}

extern "C" int rk_fasta_load_finish(rk_fasta_load* L, uint64_t total_bytes, rk_fasta_index* out) {
    if (!L || !out || total_bytes < 1 || total_bytes > L->cap || L->finished) return fail(RK_ERR_ARG, "bad arguments");
    memset(out, 0, sizeof *out);
    rk_ctx* c = L->c;
    RKCHK(set_dev(c));
    hipStream_t st = c->st;
    const uint64_t chunks = fa_chunks(total_bytes);
    FaDev& d = L->d;
    d.chunk_map = L->d_u32.as<uint32_t>(); d.chunk_pre = d.chunk_map + chunks;
    d.chunk_kept = L->d_u64.as<uint64_t>(); d.chunk_hdrs = d.chunk_kept + (chunks + 1);
    d.kept_base = d.chunk_hdrs + (chunks + 1); d.hdr_base = d.kept_base + (chunks + 1);
    d.info = L->d_info.as<uint32_t>();
    RKCHK(L->d_scan.reserve(fa_scan_temp_bytes(chunks + 1)));
    d.scan_tmp = L->d_scan.p; d.scan_tmp_bytes = L->d_scan.cap;
    const uint8_t* raw = L->d_text.as<uint8_t>();
    HIPCHK(launch_fasta_count(d, raw, total_bytes, st));
    uint64_t* hs = L->h_small.as<uint64_t>(); // [0] bases, [1] records, [2] status word, [3] name bytes, [4] offset of the first record
    HIPCHK(hipMemcpyAsync(hs, d.kept_base + chunks, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 1, d.hdr_base + chunks, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2, d.info, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t total = hs[0], nrec = hs[1];
    uint32_t status = (uint32_t)hs[2];
    if (nrec == 0) status |= FA_BAD_EMPTY;
    if (nrec >= 0x7ffffff0ull) return fail(RK_ERR_LIMIT, "more than 2^31 reference sequences");
    if (status) { out->status = (int32_t)status; return RK_OK; }
    RKCHK(L->d_bases.reserve(total + 64));
    RKCHK(L->d_rec.reserve((4 * (nrec + 1)) * 8 + 64));
    RKCHK(L->d_scan.reserve(fa_scan_temp_bytes(nrec + 1)));
    d.scan_tmp = L->d_scan.p; d.scan_tmp_bytes = L->d_scan.cap;
    d.bases = L->d_bases.as<uint8_t>();
    d.hdr_pos = L->d_rec.as<uint64_t>(); d.rec_off = d.hdr_pos + (nrec + 1);
    d.name_len1 = d.rec_off + (nrec + 1); d.name_off = d.name_len1 + (nrec + 1);
    HIPCHK(launch_fasta_compact(d, raw, total_bytes, nrec, st));
    HIPCHK(hipMemcpyAsync(hs + 2, d.info, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 3, d.name_off + nrec, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 4, d.rec_off, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    status = (uint32_t)hs[2];
    if (hs[4] != 0) status |= FA_BAD_LEAD; // bases before the first header line
    if (status) { out->status = (int32_t)status; return RK_OK; }
    const uint64_t name_bytes = hs[3];
    RKCHK(L->d_names.reserve(name_bytes + 64));
    d.names = L->d_names.as<uint8_t>();
    HIPCHK(launch_fasta_names(d, raw, nrec, st));
    L->offsets.assign((size_t)nrec + 1, 0);
    L->name_offsets.assign((size_t)nrec + 1, 0);
    L->names.assign((size_t)name_bytes + 1, 0);
    HIPCHK(hipMemcpyAsync(L->offsets.data(), d.rec_off, nrec * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(L->name_offsets.data(), d.name_off, (nrec + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(L->names.data(), d.names, name_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    L->offsets[(size_t)nrec] = total;
    L->nseq = (int64_t)nrec;
    L->finished = true;
    // the text has done its work; the packed bases stay for rk_set_references_fasta
    L->d_text.release();
    out->nseq = L->nseq;
    out->offsets = L->offsets.data();
    out->names = L->names.data();
    out->name_offsets = L->name_offsets.data();
    return RK_OK;
}

// the packed bases (offsets[nseq] bytes, as the text spells them: not upper-cased) for callers that also want them on the host
extern "C" int rk_fasta_load_get_bases(rk_fasta_load* L, uint8_t* dst) {
    if (!L || !dst || !L->finished) return fail(RK_ERR_ARG, "rk_fasta_load_get_bases needs a finished, regular rk_fasta_load");
    RKCHK(set_dev(L->c));
    const uint64_t total = L->offsets.back();
    if (total) HIPCHK(hipMemcpyAsync(dst, L->d_bases.p, total, hipMemcpyDeviceToHost, L->c->st));
    HIPCHK(hipStreamSynchronize(L->c->st));
    return RK_OK;
}

extern "C" int rk_set_references_fasta(rk_ctx* c, rk_fasta_load* L, const int* ks, int nks, int S, int max_samples, uint64_t counter_slots) {
    if (!c || !L || !L->finished) return fail(RK_ERR_ARG, "rk_set_references_fasta needs a finished, regular rk_fasta_load");
    if (L->c != c) return fail(RK_ERR_ARG, "the text was loaded through another context");
    if (L->nseq > 0x7fffffffll) return fail(RK_ERR_LIMIT, "too many reference sequences");
    return set_references_impl(c, nullptr, L->d_bases.as<uint8_t>(), L->offsets.data(), (int)L->nseq, ks, nks, S, max_samples, counter_slots);
}
