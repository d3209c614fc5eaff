// rk_counters.hip -- HASHTCounter behind the C ABI (ctor src/rkmh.cpp:739,742,1187; increment :335; get :1218,1260): the full table
// of int32 slots in HBM, the compact depth map of -M runs, their sums across devices and their (de)serialisation.
#include "rk_api_internal.hpp"

// every other reader / writer of a table first waits for the count passes enqueued so far
int counter_settle(const rk_counter* k) {
    if (k && k->last_set) HIPCHK(hipEventSynchronize(k->last));
    if (k && k->last_atomic_set) HIPCHK(hipEventSynchronize(k->last_atomic));
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

