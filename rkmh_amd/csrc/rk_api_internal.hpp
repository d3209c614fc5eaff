// rk_api_internal.hpp -- what the translation units behind the C ABI (include/rkmh_amd.h) share: the context and counter objects, the
// growable buffers, the error plumbing and the handful of internal entry points one part needs from another.
//   rk_api.hip       contexts, errors, the general path (any length), the one-sequence mirrors of the mkmh calls, the classify routing
//   rk_index.hip     reference sketches -> the resident index (buckets, postings, k-mer-space structures), depth filter masks
//   rk_counters.hip  HASHTCounter (full and compact), its (de)serialisation
//   rk_frontend.hip  FASTQ slots (text parsed on the device), BGZF jobs inflated on the device, reference FASTA through the device
//   rk_call.hip      `call`
#pragma once
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

struct rk_gzip; // rk_gunzip.hip
namespace rk {
// sets the thread's error text (rk_last_error) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}
using namespace rk;

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(RK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define RKCHK(expr) do { int _r = (expr); if (_r != RK_OK) return _r; } while (0)

// growable device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool view = false; // p lies inside another allocation (GzScratch's arena): never freed from here
    void set_view(void* at, size_t bytes) { release(); p = at; cap = bytes; view = true; }
    int reserve(size_t bytes) {
        if (bytes <= cap) return RK_OK;
        if (view) { p = nullptr; cap = 0; view = false; }
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(RK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        cap = want;
        return RK_OK;
    }
    void release() { if (p && !view) { hipError_t e = hipFree(p); (void)e; } p = nullptr; cap = 0; view = false; }
    template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool view = false;
    void set_view(void* at, size_t bytes) { release(); p = at; cap = bytes; view = true; }
    int reserve(size_t bytes) {
        if (bytes <= cap) return RK_OK;
        if (view) { p = nullptr; cap = 0; view = false; }
        if (p) { hipError_t e = hipHostFree(p); (void)e; p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc(&p, bytes + 256, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return fail(RK_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        cap = bytes + 256;
        return RK_OK;
    }
    void release() { if (p && !view) { hipError_t e = hipHostFree(p); (void)e; } p = nullptr; cap = 0; view = false; }
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

inline int set_dev(rk_ctx* c) { HIPCHK(hipSetDevice(c->device)); return RK_OK; }

// ---- the general path (rk_api.hip): hash tiles -> (optional) in-LDS sort / sketch / intersect, for sequences of any length
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

// ---- host-side helpers ----
// d_bases: device pointer to the batch's bases when already resident (else nullptr => upload from `bases`)
// memcpy into a pinned staging buffer with a few threads: one core copies ~14 GB/s, the link takes several times that
inline int host_threads() { // workers for the host-side copies and per-read loops (RKMH_COPY_THREADS; default: up to 8 of the CPUs granted)
    static const int nt = []() {
        const char* e = getenv("RKMH_COPY_THREADS");
        int v = e ? atoi(e) : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2));
        return v < 1 ? 1 : (v > 32 ? 32 : v);
    }();
    return nt;
}
// f(begin, end) over [0, n) in contiguous pieces on host_threads() threads (the caller's thread takes the first piece)
template <typename F>
inline void par_for(size_t n, size_t min_piece, F f) {
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
inline void par_memcpy(void* dst, const void* src, size_t n) {
    par_for(n, (size_t)4 << 20, [=](size_t lo, size_t hi) { memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
}

// ---- internal entry points shared between the parts (defined in the file named) ----
int check_ks(const int* ks, int nks, KsArr* out);                                                           // rk_api.hip
uint32_t next_pow2(uint32_t x);
void apply_depth_cfg(const rk_ctx* c, GeneralCfg& cfg);
int general_run(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases_in, const uint64_t* offsets, int64_t n, const GeneralCfg& cfg, const GeneralOut& out);
bool is_pinned_host(const void* p, size_t bytes);
int upload_staged(rk_ctx* c, void* dst, const uint8_t* src, size_t bytes, hipStream_t st);
int fused_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, void* d_out4, uint32_t max_read_len, int mode, rk_counter* count_into,
                 hipStream_t st, uint64_t total_bases = 0);
// rows the fused kernel flagged (max_id == -2; `rows` = host copy of d_out4) answered by the general kernels on the resident bases and scattered back into d_out4 AND rows; synchronises st
int reroute_flagged_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, void* d_out4, int32_t* rows, hipStream_t st);
// the work buffers of the device gunzip (rk_gunzip.hip): they belong to the SLOT that makes the calls, so a worker that reads file
// after file allocates them once (several gigabytes for a file of 600 MB of text; made and freed per file they cost more than the kernels)
struct GzScratch {
    DevBuf d_comp, d_chunks, d_scratch, d_planes, d_rings, d_heads, d_stage, d_misc;
    PinBuf h_chunks, h_misc;
    DevBuf arena;  // gzip_reserve: ONE device allocation that the eight buffers above are views of (a runtime call of this kind costs
    PinBuf harena; // ~40 ms while other workers are busy, whatever its size: two calls instead of ten)
    void release() {
        for (DevBuf* b : {&d_comp, &d_chunks, &d_scratch, &d_planes, &d_rings, &d_heads, &d_stage, &d_misc}) b->release();
        h_chunks.release(); h_misc.release();
        arena.release(); harena.release();
    }
};
int gzip_reserve(GzScratch& S, rk_ctx* c, uint64_t comp_bytes, uint64_t cap_out); // rk_gunzip.hip
int gzip_next(rk_gzip* gz, GzScratch& S, rk_ctx* c, hipStream_t st, hipEvent_t ev, int64_t call, uint8_t* d_out, uint64_t cap_out, uint64_t* nbytes, uint64_t* text_off, bool raw = false); // rk_gunzip.hip
int counter_settle(const rk_counter* k);                                                                    // rk_counters.hip
int build_index(rk_ctx* c);                                                                                 // rk_index.hip
int set_references_impl(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases, const uint64_t* offsets, int nref, const int* ks, int nks, int S,
                        int max_samples, uint64_t counter_slots);
