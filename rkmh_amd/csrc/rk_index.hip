// rk_index.hip -- the reference side of main_stream (src/rkmh.cpp:783-785, :816-838) behind the C ABI: sketches -> the resident
// lookup index (fingerprint buckets, posting lists of genome families, the k-mer-space filter and maps with their enumeration
// cache), and the per-key masks of the -M depth filter.
#include "rk_api_internal.hpp"

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
    // Are the bases worth their price?  The kernel that knows (base, exceptions) lists carries per-read base counters, a test of every
    // compound value and an expansion step for EVERY read of every batch (k_classify_kmer<..., FAM>: 313.7 against 304.1 us per 1 M
    // reads on BASELINE config 2's panel, whose 182 HPV types share a handful of conserved k-mers and nothing else).  They stay only
    // where they take a real share of the posting walks away: at least 5 % of all postings of the index (config 3's panel: 70 %).
    if (!bases.empty()) {
        uint64_t total = 0, saved = 0;
        for (auto& d : dl) {
            total += (uint64_t)d.weight * d.e.size();
            if (!d.simple || d.e.size() < 8) continue;
            size_t bd = ~(size_t)0;
            for (auto& b : bases) bd = std::min(bd, sym_diff(d.e, b));
            if (2 * (1 + bd) <= d.e.size()) saved += (uint64_t)d.weight * (d.e.size() - bd - 1);
        }
        if (saved * 20 < total) bases.clear();
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
        if (nex <= 8 && small_refs) {
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
int build_index(rk_ctx* c) {
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
    const size_t load_pct = 250;
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
            std::stable_sort(grp.begin(), grp.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return (a.first & 3u) < (b.first & 3u); });
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
    // keys: 1 MB 2.28 ms, 2 MB 1.77, 4 MB 1.99).  (RKMH_PRE_MAXKB overrides the cap: tests.)
    c->ix.pre = nullptr; c->ix.pmask = 0;
    int pre_mode = 1;
    if (const char* e = getenv("RKMH_PREFILTER")) pre_mode = atoi(e);
    if (pre_mode > 0) {
        size_t bits_per_key = 32, max_words = (size_t)(distinct > 2000000 ? 2048 : 1024) * 256;
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
    const size_t kmer_max_keys = 6000000;
    bool all_k_ok = kmer_mode > 0 && c->kmer_form_allowed && c->ks.n >= 1 && c->ks.n <= KM_MAX_KS && distinct <= kmer_max_keys;
    // one k of 17 .. 20 (wide k-mers, 64-bit): the 4^k enumeration takes 0.1 s (k = 17), 0.4 s (18), 1.7 s (19), 6.7 s (20) -- done unasked
    // up to RKMH_KMER_ENUM_MAXK (default 18); beyond that only with a cache file (rk_set_kmer_cache): from it, or -- once -- into it
    const int enum_maxk = getenv("RKMH_KMER_ENUM_MAXK") ? atoi(getenv("RKMH_KMER_ENUM_MAXK")) : 18; // (read per build: a few per process)
    const bool wide_k = c->ks.n == 1 && c->ks.k[0] > 16 && c->ks.k[0] <= KW_MAX_K;
    for (int j = 0; j < c->ks.n; ++j) all_k_ok = all_k_ok && c->ks.k[j] >= KPRE_MIN_K && (c->ks.k[j] <= 16 || wide_k);
    if (wide_k && distinct >= (size_t)KW_EMPTY - 16) all_k_ok = false; // (key numbers of the wide map are 20 bits)
    for (int j = 0; j + 1 < c->ks.n; ++j) for (int i = j + 1; i < c->ks.n; ++i) all_k_ok = all_k_ok && c->ks.k[j] != c->ks.k[i]; // a size given twice hashes twice: hash-space path
    clk.tick("index + prefilter uploaded");
    std::vector<uint32_t> kpost, kbase;
    std::unordered_map<uint32_t, KList> kremap;
    c->ix.kpost = nullptr; c->ix.kbase = nullptr; c->ix.kbase_n = 0; c->ix.kkeys = nullptr; c->ix.kslots = nullptr;
    if (all_k_ok) {
        build_kpost(post, R, kpost, kbase, kremap);
        if (kpost.size() >= 0x3fffffffull) all_k_ok = false;
        else {
            RKCHK(c->d_kpost.reserve(kpost.size() * 4));
            RKCHK(c->d_kbase.reserve(kbase.size() * 4));
            HIPCHK(hipMemcpy(c->d_kpost.p, kpost.data(), kpost.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(c->d_kbase.p, kbase.data(), kbase.size() * 4, hipMemcpyHostToDevice));
            c->ix.kpost = c->d_kpost.as<uint32_t>(); c->ix.kbase = c->d_kbase.as<uint32_t>();
            for (int b = 0; b < KBASE_MAX; ++b) c->ix.kbase_n += kbase[2 * (size_t)b + 1] != 0u ? 1u : 0u;
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
    if (all_k_ok && wide_k && c->ks.k[0] > enum_maxk && kcache.find(c->ks.k[0]) == kcache.end() && c->kmer_cache_path.empty()) all_k_ok = false; // too long to do for one run
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
            const double km2_load = 0.65;
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
                const double km1_load_est = 0.65;
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
                const double km1_load = 0.65;
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
int set_references_impl(rk_ctx* c, const uint8_t* bases, const uint8_t* d_bases, const uint64_t* offsets, int nref,
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

