/*
 * rk_oracle.c -- CPU ORACLE (test infrastructure only; see rk_oracle.h header).
 * PARITY UNPINNED (mkmh submodule absent from /root/reference; see rk_oracle.h).
 *
 * Plain C restatement of the rkmh classify/stream hot path.  Each function cites the
 * reference call site / in-tree analogue it follows (paths relative to /root/reference).
 */
#include "rk_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void rko_default_policy(rko_policy* p) {
    p->fold = RKO_FOLD_SWAP32;
    p->drop_last_window = 1;
    p->counter_counts_zero = 1;
    p->mask_strict_less = 1;
    p->freq_max_inclusive = 1;
    p->seed = 42; /* src/rkmh.cpp:497 ("hashSeed", 42) */
}

/* ---- MurmurHash3_x64_128 (public algorithm; hashType named at src/rkmh.cpp:495) ---- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33; return k;
}
void rko_murmur3_x64_128(const void* key, int len, uint32_t seed, void* out) {
    const uint8_t* data = (const uint8_t*)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (int i = 0; i < nblocks; i++) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t* tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
        case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
        case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
        case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
        case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
        case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
        case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
        case 9:  k2 ^= (uint64_t)tail[8];
                 k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
        case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
        case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
        case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
        case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
        case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
        case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
        case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
        case 1:  k1 ^= (uint64_t)tail[0];
                 k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    ((uint64_t*)out)[0] = h1;
    ((uint64_t*)out)[1] = h2;
}

/* U1 */
static inline uint64_t fold128(const uint32_t w[4], int fold) {
    switch (fold) {
        case RKO_FOLD_H1:   return ((uint64_t)w[1] << 32) | w[0];
        case RKO_FOLD_W2W1: return ((uint64_t)w[2] << 32) | w[1];
        default:            return ((uint64_t)w[0] << 32) | w[1];
    }
}

/* A2: recollected mkmh::to_upper: seq[i] = ((c - 91) > 0 ? c - 32 : c) on (signed) char. */
void rko_to_upper(char* s, int len) {
    for (int i = 0; i < len; i++) {
        signed char c = (signed char)s[i];
        s[i] = (char)(((int)c - 91) > 0 ? c - 32 : c);
    }
}

/* U4: a window is hashable iff every byte is upper-case A/C/G/T */
static inline int is_acgt(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
/* reverse complement over the 26-letter table: A<->T, C<->G, everything else maps to itself */
static inline char comp_base(char c) {
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; }
}

static uint64_t canonical_hash(const char* kmer, int k, char* rcbuf, const rko_policy* p) {
    for (int i = 0; i < k; i++) if (!is_acgt(kmer[i])) return 0; /* sentinel: rkmh.cpp:1218,1233 */
    for (int i = 0; i < k; i++) rcbuf[i] = comp_base(kmer[k - 1 - i]);
    uint32_t fw[4], rw[4];
    rko_murmur3_x64_128(kmer, k, p->seed, fw);
    rko_murmur3_x64_128(rcbuf, k, p->seed, rw);
    uint64_t f = fold128(fw, p->fold), r = fold128(rw, p->fold);
    return f < r ? f : r; /* U2: min over folded values */
}

uint64_t rko_calc_hash(const char* kmer, int k, const rko_policy* p) {
    char* rc = (char*)malloc((size_t)k + 1);
    uint64_t h = canonical_hash(kmer, k, rc, p);
    free(rc);
    return h;
}

int rko_num_windows(int len, int k, const rko_policy* p) {
    int n = p->drop_last_window ? len - k : len - k + 1;
    return n > 0 ? n : 0; /* unguarded in the reference; defined as empty */
}

void rko_calc_hashes(const char* seq, int len, const int* ks, int nks,
                     uint64_t** out, int* n, const rko_policy* p) {
    int total = 0;
    for (int j = 0; j < nks; j++) total += rko_num_windows(len, ks[j], p);
    uint64_t* h = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(total > 0 ? total : 1));
    int pos = 0;
    for (int j = 0; j < nks; j++) {
        int k = ks[j];
        int nw = rko_num_windows(len, k, p);
        char* rc = (char*)malloc((size_t)k + 1);
        for (int i = 0; i < nw; i++) h[pos++] = canonical_hash(seq + i, k, rc, p);
        free(rc);
    }
    *out = h; *n = total;
}

static int cmp_u64(const void* a, const void* b);

rko_counter* rko_counter_new(uint64_t slots) {
    rko_counter* c = (rko_counter*)malloc(sizeof(rko_counter));
    c->slots = slots;
    c->counts = (int32_t*)calloc(slots, sizeof(int32_t));
    return c;
}
void rko_counter_free(rko_counter* c) { if (c) { free(c->counts); free(c); } }
void rko_counter_increment(rko_counter* c, uint64_t key) {
#pragma omp atomic update
    ++c->counts[key % c->slots];
}
int32_t rko_counter_get(const rko_counter* c, uint64_t key) { return c->counts[key % c->slots]; }

void rko_calc_hashes_counted(const char* seq, int len, const int* ks, int nks,
                             uint64_t** out, int* n, rko_counter* c, const rko_policy* p) {
    rko_calc_hashes(seq, len, ks, nks, out, n, p);
    for (int i = 0; i < *n; i++)
        if (p->counter_counts_zero || (*out)[i] != 0) rko_counter_increment(c, (*out)[i]);
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* A4 (no dedup: in-tree analogue src/rkmh.cpp:1210,1233-1239) */
void rko_minhashes(uint64_t* h, int n, int S, uint64_t** mins, int* m) {
    uint64_t* r = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(S > 0 ? S : 1));
    int cnt = 0;
    if (n > 0) qsort(h, (size_t)n, sizeof(uint64_t), cmp_u64);
    for (int i = 0; i < n && cnt < S; i++)
        if (h[i] != 0) r[cnt++] = h[i];
    *mins = r; *m = cnt;
}

void rko_mask_by_frequency(uint64_t* h, int n, const rko_counter* c, int min_occ, const rko_policy* p) {
    for (int i = 0; i < n; i++) {
        int32_t v = rko_counter_get(c, h[i]);
        if (p->mask_strict_less ? (v < min_occ) : (v <= min_occ)) h[i] = 0;
    }
}

void rko_minhashes_frequency_filter(uint64_t* h, int n, int S, uint64_t** out, int* m,
                                    const rko_counter* c, int min_c, int max_c, const rko_policy* p) {
    uint64_t* r = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(S > 0 ? S : 1));
    int cnt = 0;
    if (n > 0) qsort(h, (size_t)n, sizeof(uint64_t), cmp_u64);
    for (int i = 0; i < n && cnt < S; i++) {
        if (h[i] == 0) continue;
        int32_t v = rko_counter_get(c, h[i]);
        int keep = p->freq_max_inclusive ? (v >= min_c && v <= max_c) : (v >= min_c && v < max_c);
        if (keep) r[cnt++] = h[i]; /* analogue: curr != 0 && get(curr) <= max_samples, rkmh.cpp:1218 */
    }
    *out = r; *m = cnt;
}

/* A5 */
void rko_hash_intersection_size(const uint64_t* a, int na, const uint64_t* b, int nb, int* out) {
    int i = 0, j = 0, r = 0;
    while (i < na && a[i] == 0) i++;
    while (j < nb && b[j] == 0) j++;
    while (i < na && j < nb) {
        if (a[i] == b[j]) { r++; i++; j++; }
        else if (a[i] > b[j]) j++;
        else i++;
    }
    *out = r;
}

/* A5f: equiv.hpp:308,340,364 (the callers read get<1> as the count and delete[] get<0>) */
void rko_hash_intersection(const uint64_t* a, int a_start, int a_len, const uint64_t* b, int b_start, int b_len, int S,
                           uint64_t** out, int* n) {
    const uint64_t* x = a + a_start;
    const uint64_t* y = b + b_start;
    int cap = S < a_len ? S : a_len;
    uint64_t* r = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(cap > 0 ? cap : 1));
    int i = 0, j = 0, c = 0;
    while (i < a_len && x[i] == 0) i++;
    while (j < b_len && y[j] == 0) j++;
    while (i < a_len && j < b_len && c < cap) {
        if (x[i] == y[j]) { r[c++] = x[i]; i++; j++; }
        else if (x[i] > y[j]) j++;
        else i++;
    }
    *out = r; *n = c;
}

/* A6: src/rkmh.cpp:874-883 */
void rko_argmax_diff(const int* shared, int R, int* max_id, int* max_shared, int* diff) {
    int ms = -1, mi = 0, d = 0;
    for (int j = 0; j < R; j++) {
        if (shared[j] > ms) { d = shared[j] - ms; ms = shared[j]; mi = j; }
    }
    *max_id = mi; *max_shared = ms; *diff = d;
}

void rko_free(void* p) { free(p); }

int rko_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static char* upper_copy(const char* s, int len) {
    char* x = (char*)malloc((size_t)len + 1);
    memcpy(x, s, (size_t)len);
    rko_to_upper(x, len);
    return x;
}

/* A1: src/rkmh.cpp:816-826 */
void rko_sketch_refs(const char* bases, const uint64_t* offsets, int nref,
                     const int* ks, int nks, int S,
                     uint64_t* sketches, int32_t* sketch_lens, const rko_policy* p, int threads) {
    (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (int i = 0; i < nref; i++) {
        int len = (int)(offsets[i + 1] - offsets[i]);
        char* s = upper_copy(bases + offsets[i], len);
        uint64_t* h; int num; uint64_t* mins; int m;
        rko_calc_hashes(s, len, ks, nks, &h, &num, p);
        rko_minhashes(h, num, S, &mins, &m);
        memcpy(sketches + (size_t)i * S, mins, sizeof(uint64_t) * (size_t)m);
        for (int j = m; j < S; j++) sketches[(size_t)i * S + j] = 0;
        sketch_lens[i] = m;
        free(h); free(mins); free(s);
    }
}

/* -I path: src/rkmh.cpp:828-838 (counter incremented per k-mer occurrence: SURVEY Appendix C.7) */
void rko_sketch_refs_maxsamples(const char* bases, const uint64_t* offsets, int nref,
                                const int* ks, int nks, int S, int max_samples, uint64_t counter_slots, int distinct,
                                uint64_t* sketches, int32_t* sketch_lens, const rko_policy* p, int threads) {
    (void)threads;
    rko_counter* c = rko_counter_new(counter_slots);
    uint64_t** hs = (uint64_t**)malloc(sizeof(uint64_t*) * (size_t)(nref > 0 ? nref : 1));
    int* ns = (int*)malloc(sizeof(int) * (size_t)(nref > 0 ? nref : 1));
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (int i = 0; i < nref; i++) {
        int len = (int)(offsets[i + 1] - offsets[i]);
        /* refs were upper-cased by parse_fastas (rkmh.cpp:252) */
        char* s = upper_copy(bases + offsets[i], len);
        if (!distinct) rko_calc_hashes_counted(s, len, ks, nks, &hs[i], &ns[i], c, p);
        else {
            /* main_filter's hash_sequences, src/rkmh.cpp:343-355: std::set of the sample's hashes, one increment per member */
            rko_calc_hashes(s, len, ks, nks, &hs[i], &ns[i], p);
            uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(ns[i] > 0 ? ns[i] : 1));
            memcpy(tmp, hs[i], sizeof(uint64_t) * (size_t)ns[i]);
            if (ns[i] > 0) qsort(tmp, (size_t)ns[i], sizeof(uint64_t), cmp_u64);
            for (int j = 0; j < ns[i]; j++)
                if (j == 0 || tmp[j] != tmp[j - 1]) rko_counter_increment(c, tmp[j]);
            free(tmp);
        }
        free(s);
    }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic)
    for (int i = 0; i < nref; i++) {
        uint64_t* mins; int m;
        rko_minhashes_frequency_filter(hs[i], ns[i], S, &mins, &m, c, 0, max_samples, p);
        memcpy(sketches + (size_t)i * S, mins, sizeof(uint64_t) * (size_t)m);
        for (int j = m; j < S; j++) sketches[(size_t)i * S + j] = 0;
        sketch_lens[i] = m;
        free(mins); free(hs[i]);
    }
    free(hs); free(ns);
    rko_counter_free(c);
}

/* A0: src/rkmh.cpp:845-898 restated literally: to_upper, calc_hashes, minhashes,
 * hash_intersection_size against EVERY ref by two-pointer merge, sequential argmax. */
void rko_classify_stream(const char* bases, const uint64_t* offsets, int64_t nreads,
                         const int* ks, int nks, int S,
                         const uint64_t* ref_sketches, const int32_t* ref_lens, int nref,
                         int32_t* out4, const rko_policy* p, int threads) {
    (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < nreads; i++) {
        int len = (int)(offsets[i + 1] - offsets[i]);
        int* shared_arr = (int*)malloc(sizeof(int) * (size_t)(nref > 0 ? nref : 1));
        char* s = upper_copy(bases + offsets[i], len);             /* :856 */
        uint64_t* h; int num; uint64_t* mins; int min_num;
        rko_calc_hashes(s, len, ks, nks, &h, &num, p);             /* :860 */
        rko_minhashes(h, num, S, &mins, &min_num);                 /* :863 */
        free(h); free(s);
        for (int j = 0; j < nref; j++)                             /* :867-870 */
            rko_hash_intersection_size(mins, min_num, ref_sketches + (size_t)j * S, ref_lens[j], &shared_arr[j]);
        int max_id, max_shared, diff;
        rko_argmax_diff(shared_arr, nref, &max_id, &max_shared, &diff); /* :874-883 */
        out4[i * 4 + 0] = max_id; out4[i * 4 + 1] = max_shared;
        out4[i * 4 + 2] = diff;   out4[i * 4 + 3] = min_num;
        free(mins); free(shared_arr);
    }
}

/* A0': src/rkmh.cpp:904-948 */
void rko_classify_stream_depth(const char* bases, const uint64_t* offsets, int64_t nreads,
                               const int* ks, int nks, int S,
                               const uint64_t* ref_sketches, const int32_t* ref_lens, int nref,
                               int min_kmer_occ, uint64_t counter_slots,
                               int32_t* out4, const rko_policy* p, int threads) {
    (void)threads;
    rko_counter* c = rko_counter_new(counter_slots);
    uint64_t** hs = (uint64_t**)malloc(sizeof(uint64_t*) * (size_t)(nreads > 0 ? nreads : 1));
    int* ns = (int*)malloc(sizeof(int) * (size_t)(nreads > 0 ? nreads : 1));
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < nreads; i++) {                          /* pass 1 :904-910 */
        int len = (int)(offsets[i + 1] - offsets[i]);
        char* s = upper_copy(bases + offsets[i], len);
        rko_calc_hashes_counted(s, len, ks, nks, &hs[i], &ns[i], c, p);
        free(s);
    }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < nreads; i++) {                          /* pass 2 :911-948 */
        int* shared_arr = (int*)malloc(sizeof(int) * (size_t)(nref > 0 ? nref : 1));
        uint64_t* mins; int num_mins;
        rko_mask_by_frequency(hs[i], ns[i], c, min_kmer_occ, p);    /* :916 */
        rko_minhashes(hs[i], ns[i], S, &mins, &num_mins);           /* :917 */
        free(hs[i]);
        for (int j = 0; j < nref; j++)
            rko_hash_intersection_size(mins, num_mins, ref_sketches + (size_t)j * S, ref_lens[j], &shared_arr[j]);
        int max_id, max_shared, diff;
        rko_argmax_diff(shared_arr, nref, &max_id, &max_shared, &diff);
        out4[i * 4 + 0] = max_id; out4[i * 4 + 1] = max_shared;
        out4[i * 4 + 2] = diff;   out4[i * 4 + 3] = num_mins;
        free(mins); free(shared_arr);
    }
    free(hs); free(ns);
    rko_counter_free(c);
}
