/*
 * rk_oracle.h -- CPU ORACLE for the rkmh classify/stream hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (rkmh_amd/) never links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the arithmetic of the hot path lives in the un-vendored
 * submodule github.com/edawson/mkmh (+ mkmh/murmur3), whose directory is empty
 * in /root/reference and whose pinned commit is unknown (.gitmodules:1-3).  The
 * reference has no tests or golden vectors.  Everything below is
 * therefore a restatement of (i) the in-tree call sites and in-tree analogues
 * in src/rkmh.cpp / src/equiv.hpp (cited per function), (ii) the public
 * MurmurHash3_x64_128 algorithm (Appleby, public domain) which IS pinned by
 * known-answer vectors (tests/golden/murmur3_kat.json), and (iii) the
 * recollected mkmh behaviour, with every unpinned choice
 * U1..U12 isolated behind the rko_policy struct.
 */
#ifndef RK_ORACLE_H
#define RK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* U1: how the 128-bit murmur output (viewed as uint32_t w[4]) becomes a 64-bit hash_t */
enum {
    RKO_FOLD_SWAP32 = 0, /* ((uint64_t)w[0] << 32) | w[1]   (recollected mkmh HEAD; default) */
    RKO_FOLD_H1     = 1, /* *(uint64_t*)w = h1               (Mash-compatible)                 */
    RKO_FOLD_W2W1   = 2  /* ((uint64_t)w[2] << 32) | w[1]   (recollected 2016 mkmh.cpp)       */
};

typedef struct rko_policy {
    int32_t fold;             /* U1  */
    int32_t drop_last_window; /* U3: 1 => numhashes = len-k (default), 0 => len-k+1 */
    int32_t counter_counts_zero; /* U12: 1 => 6-arg calc_hashes increments the 0 sentinel too (default) */
    int32_t mask_strict_less; /* U9: 1 => zero a hash when count <  min_occ (default); 0 => <= */
    int32_t freq_max_inclusive; /* U10: 1 => keep when min <= count <= max (default) */
    uint32_t seed;            /* 42: rkmh.cpp:497 hashSeed */
} rko_policy;

void rko_default_policy(rko_policy* p);

/* Public-domain MurmurHash3_x64_128; out = 16 bytes (h1 then h2, native little endian). */
void rko_murmur3_x64_128(const void* key, int len, uint32_t seed, void* out);

/* A2: to_upper call sites rkmh.cpp:227,818,856,908 (mkmh quirk: every byte > 91 gets -32). */
void rko_to_upper(char* s, int len);

/* A3s: single canonical k-mer hash, call sites rkmh.cpp:1811,1852 */
uint64_t rko_calc_hash(const char* kmer, int k, const rko_policy* p);

/* number of windows for one k (U3) */
int rko_num_windows(int len, int k, const rko_policy* p);

/* A3: calc_hashes(seq,len,kmer-vector,out,n) call sites rkmh.cpp:821,860.
 * Allocates *out with malloc (caller frees with rko_free). Multi-k = concatenation in k order (U5). */
void rko_calc_hashes(const char* seq, int len, const int* ks, int nks,
                     uint64_t** out, int* n, const rko_policy* p);

/* A9: HASHTCounter, ctor sizes rkmh.cpp:739,742; slot = key % slots, int32 (U8) */
typedef struct rko_counter { uint64_t slots; int32_t* counts; } rko_counter;
rko_counter* rko_counter_new(uint64_t slots);
void rko_counter_free(rko_counter* c);
void rko_counter_increment(rko_counter* c, uint64_t key);
int32_t rko_counter_get(const rko_counter* c, uint64_t key);

/* A3': 6-arg calc_hashes (rkmh.cpp:831,909): A3 + counter increment per hash */
void rko_calc_hashes_counted(const char* seq, int len, const int* ks, int nks,
                             uint64_t** out, int* n, rko_counter* c, const rko_policy* p);

/* A4: minhashes(h,n,S,mins,m) call sites rkmh.cpp:822,863; in-tree analogue rkmh.cpp:1210,1233-1239.
 * Sorts h IN PLACE ascending, skips zeros, copies the first <= S. *mins malloc'd. */
void rko_minhashes(uint64_t* h, int n, int S, uint64_t** mins, int* m);

/* A8: mask_by_frequency (rkmh.cpp:916) */
void rko_mask_by_frequency(uint64_t* h, int n, const rko_counter* c, int min_occ, const rko_policy* p);

/* A8r: minhashes_frequency_filter (rkmh.cpp:835-836; analogue rkmh.cpp:1218) */
void rko_minhashes_frequency_filter(uint64_t* h, int n, int S, uint64_t** out, int* m,
                                    const rko_counter* c, int min_c, int max_c, const rko_policy* p);

/* A5: hash_intersection_size (rkmh.cpp:869,922): two-pointer merge, both advance on equality (U7) */
void rko_hash_intersection_size(const uint64_t* a, int na, const uint64_t* b, int nb, int* out);

/* A5f: the 7-argument hash_intersection filter's helpers call (equiv.hpp:308,340,364; argument order as CALLED: array,
 * start, length -- U7b): the same merge, materialising the matches (ascending, at most S); *out is malloc'd. */
void rko_hash_intersection(const uint64_t* a, int a_start, int a_len, const uint64_t* b, int b_start, int b_len, int S,
                           uint64_t** out, int* n);

/* A6: argmax + diff exactly as the sequential scan rkmh.cpp:874-883 */
void rko_argmax_diff(const int* shared, int R, int* max_id, int* max_shared, int* diff);

void rko_free(void* p);

/* A1: ref sketch build rkmh.cpp:816-826 for a batch of refs (concatenated bases + nref+1 offsets).
 * sketches: caller-provided [nref * S] u64, sketch_lens: [nref]. Upper-cases a private copy. */
void rko_sketch_refs(const char* bases, const uint64_t* offsets, int nref,
                     const int* ks, int nks, int S,
                     uint64_t* sketches, int32_t* sketch_lens, const rko_policy* p, int threads);

/* A0: the literal per-read loop rkmh.cpp:845-898 (no -M), OpenMP over reads.
 * out4[i*4+0..3] = max_id, max_shared, diff, min_num. */
void rko_classify_stream(const char* bases, const uint64_t* offsets, int64_t nreads,
                         const int* ks, int nks, int S,
                         const uint64_t* ref_sketches, const int32_t* ref_lens, int nref,
                         int32_t* out4, const rko_policy* p, int threads);

/* A0': the -M two-pass loop rkmh.cpp:904-948 (counter slots = rkmh.cpp:739 unless overridden) */
void rko_classify_stream_depth(const char* bases, const uint64_t* offsets, int64_t nreads,
                               const int* ks, int nks, int S,
                               const uint64_t* ref_sketches, const int32_t* ref_lens, int nref,
                               int min_kmer_occ, uint64_t counter_slots,
                               int32_t* out4, const rko_policy* p, int threads);

/* -I ref path rkmh.cpp:828-838 (distinct = 0: counter per k-mer occurrence) and main_filter's variant
 * rkmh.cpp:343-355 + 1211-1231 (distinct = 1: once per distinct hash per reference) */
void rko_sketch_refs_maxsamples(const char* bases, const uint64_t* offsets, int nref,
                                const int* ks, int nks, int S, int max_samples, uint64_t counter_slots, int distinct,
                                uint64_t* sketches, int32_t* sketch_lens, const rko_policy* p, int threads);

int rko_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
