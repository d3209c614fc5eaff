// rk_api.hip -- the C ABI of include/rkmh_amd.h on top of the gfx950 kernels (rk_kernels.hip).
// Host-side orchestration only: device memory, tile descriptors, the reference index build, the
// pinned double-buffered H2D/D2H pipeline.  No CPU implementation of any hashing/sketching step lives
// here: every entry point fails with RK_ERR_HIP when no GPU is usable.
#include "rk_api_internal.hpp"

static thread_local std::string g_err;
int rk::fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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

int check_ks(const int* ks, int nks, KsArr* out) {
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
uint32_t next_pow2(uint32_t x) { uint32_t p = 64; while (p < x) p <<= 1; return p; }

// mask_by_frequency of the general path: by slot of the depth table, or -- compact depth map -- through the keep bits of the index keys
void apply_depth_cfg(const rk_ctx* c, GeneralCfg& cfg) {
    if (!c->depth) return;
    if (c->depth->compact) { cfg.filter_mode = FILTER_KEYMASK; return; }
    cfg.filt_counter = c->depth; cfg.filter_mode = FILTER_MASK_MIN; cfg.fmin = c->min_occ;
}

// is [p, p + bytes) page-locked host memory the DMA engines can read directly (rk_host_alloc, hipHostMalloc, hipHostRegister)?
bool is_pinned_host(const void* p, size_t bytes) {
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
int upload_staged(rk_ctx* c, void* dst, const uint8_t* src, size_t bytes, hipStream_t st) {
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

int general_run(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases_in, const uint64_t* offsets, int64_t n,
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

int fused_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, void* d_out4,
                 uint32_t max_read_len, int mode, rk_counter* count_into, hipStream_t st, uint64_t total_bases) {
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

// Rows the fused kernel handed back (max_id == -2 in `rows`, the host copy of d_out4: long reads, reads with more windows than the
// sketch keeps, ...) answered by the general kernels on the RESIDENT bases -- only the 4-byte offsets and the flagged rows cross
// the link -- and written into rows and d_out4.  Synchronises st.
int reroute_flagged_device(rk_ctx* c, const void* d_bases, const void* d_offs, int64_t nreads, void* d_out4, int32_t* rows, hipStream_t st) {
    std::vector<uint32_t> idx;
    for (int64_t i = 0; i < nreads; ++i) if (rows[(size_t)i * 4] == -2) idx.push_back((uint32_t)i);
    if (idx.empty()) return RK_OK;
    std::vector<uint32_t> offs32((size_t)nreads + 1);
    HIPCHK(hipMemcpyAsync(offs32.data(), d_offs, ((size_t)nreads + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
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
    std::lock_guard<std::mutex> lock(c->general_mu); // (the general path works in the context's own buffers)
    RKCHK(general_run(c, nullptr, (const uint8_t*)d_bases, lens_ps.data(), (int64_t)m, cfg, go));
    for (size_t j = 0; j < m; ++j) memcpy(rows + (size_t)idx[j] * 4, res.data() + j * 4, 16);
    // scatter the answers into the device rows too
    RKCHK(c->w_ids.reserve(m * 4));
    RKCHK(c->w_out.reserve(m * 16));
    HIPCHK(hipMemcpyAsync(c->w_ids.p, idx.data(), m * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->w_out.p, res.data(), m * 16, hipMemcpyHostToDevice, st));
    HIPCHK(launch_scatter_rows(c->w_out.as<int32_t>(), c->w_ids.as<uint32_t>(), (uint32_t)m, (int32_t*)d_out4, st));
    HIPCHK(hipStreamSynchronize(st));
    return RK_OK;
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
    HIPCHK(hipMemcpyAsync(rows.data(), d_out4, (size_t)nreads * 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return reroute_flagged_device(c, d_bases, d_offs, nreads, d_out4, rows.data(), st);
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
    // chunks of 2 M / 512 k / 128 k reads): 4 M reads 278 / 292 / 252 M reads/s, 16 M reads 300 / 313 M reads/s.  (Fixed: the override this was measured with is gone.)
    // (The count pass keeps chunks of 2 M reads: its slot-partitioned form streams the whole table once per launch, rk_count.hip.)
    const int64_t MAX_READS = mode == 1 ? (int64_t)1 << 21 : (int64_t)1 << 19;
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

