/*
 * rkmh_amd.h -- C ABI of the MI355X-native rkmh classify/stream hot path.
 *
 * Drop-in boundary.  The reference has no FFI layer; the hot path sits behind
 * two concentric interfaces, both mirrored here with plain pointers and sizes:
 *
 *   inner  = the mkmh free functions called from /root/reference/src/rkmh.cpp and src/equiv.hpp
 *            (to_upper, calc_hashes, calc_hash, minhashes, mask_by_frequency,
 *             minhashes_frequency_filter, hash_intersection_size, HASHTCounter), and
 *   outer  = the per-read loop of main_stream (src/rkmh.cpp:813-898 and :904-948), which this
 *            library replaces by batched entry points (rk_set_references + rk_classify_batch*).
 *
 * Every compute entry point runs hand-written HIP kernels on a gfx950 device; there is NO CPU
 * fallback: without a usable GPU rk_ctx_create fails with RK_ERR_HIP and nothing else can be called.
 *
 * Ownership: buffers returned through `uint64_t**` are malloc'd by the library and released with
 * rk_free (the reference's convention is callee-new[] / caller-delete[], e.g. src/rkmh.cpp:823,864;
 * new[] must not cross a C ABI).  Batched entry points only use caller-owned buffers.
 * Errors: every function returns RK_OK (0) or a negative RK_ERR_*; rk_last_error() gives the text
 * (thread-local).  Threading: one rk_ctx per host thread (the reference calls mkmh concurrently from
 * OpenMP workers on disjoint data, src/rkmh.cpp:845-870; here concurrency is the batch itself).
 */
#ifndef RKMH_AMD_H
#define RKMH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_OK 0
#define RK_ERR_ARG (-1)   /* bad argument */
#define RK_ERR_HIP (-2)   /* HIP runtime / no device */
#define RK_ERR_NOMEM (-3)
#define RK_ERR_STATE (-4) /* e.g. classify before rk_set_references */
#define RK_ERR_LIMIT (-5) /* exceeds a documented limit */
#define RK_ERR_IO (-6)
#define RK_ERR_NEED_FULL (-7) /* a compact depth map cannot answer this input: repeat the pass with a full rk_counter */

#define RK_MAX_KS 8     /* number of -k values per run (src/rkmh.cpp:680-682 pushes onto a vector) */
#define RK_MAX_K 64     /* largest k-mer size */
#define RK_MAX_SKETCH 16384 /* largest -s handled by the in-LDS sorter */

/* Unpinned mkmh choices -- identical meaning to oracle/rk_oracle.h */
enum { RK_FOLD_SWAP32 = 0, RK_FOLD_H1 = 1, RK_FOLD_W2W1 = 2 };
typedef struct rk_policy {
    int32_t fold;                /* U1 */
    int32_t drop_last_window;    /* U3: 1 => len-k windows */
    int32_t counter_counts_zero; /* U12 */
    int32_t mask_strict_less;    /* U9 */
    int32_t freq_max_inclusive;  /* U10 */
    uint32_t seed;               /* 42, src/rkmh.cpp:497 */
} rk_policy;
void rk_default_policy(rk_policy* p);
/* The policy as text -- what `--hash-policy` / RKMH_POLICY of bin/rkmh and rkmh_amd.cli take and what a sketch file records
 * ("hashPolicy", next to the hashType / hashSeed keys of src/rkmh.cpp:493-497).  spec = comma-separated items applied left to
 * right onto *p (initialise it first, e.g. rk_default_policy): a preset -- `default`, or `mash`: the first 64 bits of
 * MurmurHash3_x64_128 over all len-k+1 windows, seed 42, the sketches Mash / sourmash compute (README.md:12) -- or key=value:
 * fold=swap32|h1|w2w1 (U1), windows=len-k|len-k+1 (U3), zero=count|skip (U12), mask=lt|le (U9), freqmax=incl|excl (U10), seed=<n>.
 * rk_policy_describe writes the canonical text with every key spelled out (returns its length or a negative error).
 * rk_policy_same_hashes: 1 when two policies give the same hash values and sketches (fold, window rule, seed). */
int rk_policy_parse(const char* spec, rk_policy* p);
int rk_policy_describe(const rk_policy* p, char* dst, size_t cap);
int rk_policy_same_hashes(const rk_policy* a, const rk_policy* b);

typedef struct rk_ctx rk_ctx;
typedef struct rk_counter rk_counter;

const char* rk_last_error(void);
const char* rk_version(void);
int rk_device_count(void);
/* Compute units, peak engine clock (kHz), L2 bytes (one XCD's) and HBM bytes of a device (hipGetDeviceProperties); any
 * out pointer may be NULL.  Used by bench.py to turn instruction counts into issue-slot fractions on the actual part. */
int rk_device_props(int device, int32_t* compute_units, int32_t* clock_khz, int64_t* l2_bytes, int64_t* hbm_bytes);

/* One context = one GPU (one process per GPU in multi-GPU runs). policy may be NULL (defaults). */
int rk_ctx_create(int device, const rk_policy* policy, rk_ctx** out);
int rk_ctx_policy(const rk_ctx* ctx, rk_policy* out); /* the policy the context was created with */
void rk_ctx_destroy(rk_ctx* ctx);
int rk_ctx_synchronize(rk_ctx* ctx);
/* The context's own non-blocking hipStream_t (what the host-buffer entry points enqueue on). */
void* rk_ctx_stream(rk_ctx* ctx);
void rk_free(void* p);
/* Page-locked host memory (hipHostMalloc).  rk_classify_batch / rk_count_batch read page-locked `bases` by DMA where they lie and
 * write page-locked `out4` in place -- no staging copy on either side (the reference has no analogue: its reads never leave the
 * host, src/rkmh.cpp:845-898); pageable buffers keep working through the library's own pinned staging buffers. */
int rk_host_alloc(size_t bytes, void** out);
void rk_host_free(void* p);

/* ------------------------------------------------------------------------------------------------
 * INNER boundary: one-sequence mirrors of the mkmh calls (replaces, file:line of the call site):
 * ---------------------------------------------------------------------------------------------- */
/* mkmh::to_upper(char*, int)                         src/rkmh.cpp:227,818,856,908 */
int rk_to_upper(rk_ctx* ctx, char* seq, int len);
/* mkmh::calc_hashes(const char*, int, vector<int>&, hash_t*&, int&)   src/rkmh.cpp:821,860,2101 */
int rk_calc_hashes(rk_ctx* ctx, const char* seq, int len, const int* ks, int nks, uint64_t** out, int* n);
/* 6-arg form with HASHTCounter*                      src/rkmh.cpp:831,909,1611,1616 */
int rk_calc_hashes_counted(rk_ctx* ctx, const char* seq, int len, const int* ks, int nks,
                           uint64_t** out, int* n, rk_counter* counter);
/* mkmh::calc_hash(string)                            src/rkmh.cpp:1811,1852,2198 */
int rk_calc_hash(rk_ctx* ctx, const char* kmer, int k, uint64_t* out);
/* mkmh::minhashes(hash_t*, int, int, hash_t*&, int&) src/rkmh.cpp:822,863,917 (sorts h in place) */
int rk_minhashes(rk_ctx* ctx, uint64_t* h, int n, int sketch_size, uint64_t** mins, int* m);
/* mkmh::mask_by_frequency(hash_t*, int, HASHTCounter*, int)  src/rkmh.cpp:916 */
int rk_mask_by_frequency(rk_ctx* ctx, uint64_t* h, int n, const rk_counter* counter, int min_occ);
/* mkmh::minhashes_frequency_filter(...)              src/rkmh.cpp:835-836 */
int rk_minhashes_frequency_filter(rk_ctx* ctx, uint64_t* h, int n, int sketch_size, uint64_t** out, int* m,
                                  const rk_counter* counter, int min_count, int max_count);
/* mkmh::hash_intersection_size(const hash_t*, int, const hash_t*, int, int&)  src/rkmh.cpp:869,922 */
int rk_hash_intersection_size(rk_ctx* ctx, const uint64_t* a, int na, const uint64_t* b, int nb, int* out);
/* mkmh::hash_intersection(hash_t*, int start, int len, hash_t*, int start, int len, int sketch_size) -> tuple<hash_t*, int>
 * as called from src/equiv.hpp:308,340,364 (the comment at equiv.hpp:303-305 names the arguments (ptr, len, start); the CALLS
 * pass (ptr, start, len) and that order is kept).  The matching hashes of a[a_start, a_start+a_len) and b[...], ascending,
 * both sides advancing on equality as in rk_hash_intersection_size, at most sketch_size of them.  *out is allocated by
 * the callee (the reference's callers delete[] it, equiv.hpp:317,349); release it with rk_free. */
int rk_hash_intersection(rk_ctx* ctx, const uint64_t* a, int a_start, int a_len, const uint64_t* b, int b_start, int b_len,
                         int sketch_size, uint64_t** out, int* n);

/* HASHTCounter (ctor src/rkmh.cpp:739,742,1187; increment :335; get :1218,1260): int32 slots in HBM,
 * slot = key % slots.  rk_counter_wrap adopts caller-owned DEVICE memory (e.g. a torch tensor, so
 * that the table can be all-reduced over RCCL between pass 1 and pass 2 of the -M path).  The library orders ITS passes into a
 * table among themselves; work the CALLER has in flight on the memory (a fill on torch's stream, a collective) must have
 * finished -- or be on the stream given to the pass -- before a pass is started: the passes run on the context's or the given
 * stream, not on the caller's. */
int rk_counter_create(rk_ctx* ctx, uint64_t slots, rk_counter** out);
int rk_counter_wrap(rk_ctx* ctx, void* d_counts_int32, uint64_t slots, rk_counter** out);
void rk_counter_destroy(rk_counter* c);   /* allowed after rk_ctx_destroy of its context; every other call is not */
int rk_counter_clear(rk_counter* c);
/* dst += src and dst = src, element-wise, for two tables of equal size that may belong to contexts on DIFFERENT devices: the
 * reduce / broadcast of a -M run that spreads its reads over several GPUs inside one process (bin/rkmh --devices).  The
 * reference's OpenMP threads share one HASHTCounter (src/rkmh.cpp:739, :909); one table per device summed after pass 1 gives the
 * same counts.  Both calls synchronise the two contexts' streams. */
int rk_counter_add(rk_counter* dst, const rk_counter* src);
int rk_counter_copy(rk_counter* dst, const rk_counter* src);
int rk_counter_increment(rk_counter* c, uint64_t key);
int rk_counter_get(const rk_counter* c, uint64_t key, int32_t* out);
/* Counter (de)serialisation (what the reference's commented-out read_hash_counter.write_to_binary / deserialize would do,
 * src/rkmh.cpp:744-769, :911-913; docs/todo.md:1).  File = "RKHT2\n", u64 slots, u64 nnz, u32 tag_len, tag, then
 * nnz x (u32 slot, i32 count), little endian ("RKHT1\n" files of round 1, without the tag field, load as untagged).
 * rk_counter_load* require a counter with the same number of slots and REPLACE its contents.  A file saved with a
 * provenance tag only loads through rk_counter_load_tagged with the identical tag (RK_ERR_ARG otherwise); an untagged
 * file only loads through rk_counter_load. */
int rk_counter_save(rk_counter* c, const char* path);
int rk_counter_load(rk_counter* c, const char* path);
int rk_counter_save_tagged(rk_counter* c, const char* path, const void* tag, uint32_t tag_len);
int rk_counter_load_tagged(rk_counter* c, const char* path, const void* tag, uint32_t tag_len);
/* Provenance tag of a READ depth map (pass 1 of -M, src/rkmh.cpp:904-910): k list, seed, fold, window and zero-counting
 * policy of the context, and a fingerprint of the read set (count, total bases, hashes of the lengths and of up to
 * 2 MiB of bases).  Host-side. */
#define RK_DEPTH_TAG_BYTES 128
int rk_depth_map_tag(const rk_ctx* ctx, const int* ks, int nks, const uint8_t* bases, const uint64_t* offsets,
                     int64_t nseq, uint8_t tag[RK_DEPTH_TAG_BYTES]);
void* rk_counter_device_ptr(rk_counter* c);
uint64_t rk_counter_slots(const rk_counter* c);
/* Compact depth map for -M runs that need min_num only up to bound 0 (rk_set_min_num_bound(ctx, 0): `stream` without -N,
 * `filter` with -D >= 0).  mask_by_frequency (src/rkmh.cpp:916) then changes a read's row only through windows whose hash is a
 * sketch hash, and whether such a hash survives depends on ONE slot of the HASHTCounter, hash % slots -- so only the slots that
 * some key of the reference index maps to are counted: rk_counter_entries() int32 instead of `slots` (a few hundred KB instead of
 * the 800 MB of rkmh.cpp:739), counted EXACTLY (every window of every read is still hashed; collisions into a tracked slot count,
 * as in the full table).  Laid out from the references of `ctx` at this call (create it after rk_set_references; it is refused
 * once the references change).  d_counts_int32: NULL (the library allocates and zeroes) or caller-owned, ZEROED device memory of
 * rk_counter_compact_entries() int32 (e.g. a torch tensor to all-reduce between the passes).
 * Serves rk_count_batch / rk_count_batch_device / rk_fastq_slot_count, rk_counter_clear / _add / _copy / _get (of index keys) and
 * rk_set_depth_filter.  Any batch holding a read with more hashes than the sketch keeps is refused with RK_ERR_NEED_FULL before
 * anything is counted (bottom-s selection depends on the depth of every hash): the caller repeats the pass with a full table. */
int rk_counter_create_compact(rk_ctx* ctx, uint64_t slots, void* d_counts_int32, rk_counter** out);
int rk_counter_compact_entries(const rk_ctx* ctx, uint64_t slots, uint64_t* entries);
uint64_t rk_counter_entries(const rk_counter* c); /* int32 entries behind rk_counter_device_ptr (= slots for a full table) */
int rk_counter_is_compact(const rk_counter* c);

/* ------------------------------------------------------------------------------------------------
 * OUTER boundary: batched replacements of main_stream's loops.
 * Sequences are concatenated ASCII bytes (any case; upper-casing happens on the device, as
 * src/rkmh.cpp:856 does per read) + offsets[n+1] (offsets[i]..offsets[i+1] = sequence i).
 * ---------------------------------------------------------------------------------------------- */

/* calc_hashes for a batch (hash sub-command, src/rkmh.cpp:2097-2104).  hash_offsets[n+1] (caller
 * array) receives the per-sequence segment bounds; *out is malloc'd [hash_offsets[n]] (rk_free). */
int rk_hash_batch(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int64_t nseq,
                  const int* ks, int nks, uint64_t** out, uint64_t* hash_offsets);

/* calc_hashes + minhashes for a batch (src/rkmh.cpp:816-826 for refs, :860-863 for reads).
 * sketches: caller [nseq * sketch_size] (zero padded), lens: caller [nseq]. */
int rk_sketch_batch(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int64_t nseq,
                    const int* ks, int nks, int sketch_size, uint64_t* sketches, int32_t* lens);

/* Reference side of main_stream (src/rkmh.cpp:783-785 + :816-826): sketch every reference on the
 * GPU and build the resident lookup index.  max_samples < 0 => plain path; >= 0 => the -I path
 * (src/rkmh.cpp:828-838) with a counter of counter_slots (0 => 200000000, src/rkmh.cpp:742). */
int rk_set_references(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int nref,
                      const int* ks, int nks, int sketch_size, int max_samples, uint64_t counter_slots);
/* How the -I counter is filled by the next rk_set_references: 0 = once per k-mer occurrence (main_stream,
 * src/rkmh.cpp:831), 1 = once per distinct hash per reference (main_filter's hash_sequences, src/rkmh.cpp:348-355). */
int rk_set_reference_count_mode(rk_ctx* ctx, int mode);
/* Import already-built sketches (after an RCCL broadcast from the rank that sketched them). */
int rk_set_reference_sketches(rk_ctx* ctx, const uint64_t* sketches, const int32_t* lens, int nref,
                              const int* ks, int nks, int sketch_size);
/* Export: sketches [nref*sketch_size], lens [nref] (caller arrays). */
int rk_get_reference_sketches(rk_ctx* ctx, uint64_t* sketches, int32_t* lens);
int rk_num_references(const rk_ctx* ctx);

/* Which form of the fused kernel plain classification (no -M) will use for the references now set: returns 1 when the
 * k-mer-space form is active (single k from 8 to 16: the 4^k k-mer universe was enumerated and every k-mer whose canonical hash
 * is a sketch hash -- or 0 -- is known, so windows are filtered and resolved by k-mer and never hashed), 0 for the hash-space
 * form, negative on error.  *kmers_found (may be NULL) = k-mers the enumeration found (one per strand pair). */
int rk_kmer_form(const rk_ctx* ctx, uint32_t* kmers_found);
/* Whether the NEXT rk_set_references / rk_set_reference_sketches may build the k-mer-space form (default 1).  A caller that only
 * uses the general kernels on these references (hpv16's rk_classify_groups_batch) saves the enumeration by passing 0. */
int rk_set_kmer_form(rk_ctx* ctx, int enable);
/* The k-mer-space form rests on an enumeration of the whole 4^k k-mer universe (every k-mer whose canonical hash is an index key:
 * 26 ms at k = 16 on MI355X, 4 x per further base) -- a function of the index keys, k, fold and seed alone.  With a cache file set,
 * the NEXT rk_set_references / rk_set_reference_sketches loads the lists from it when its tag (a hash of exactly those inputs)
 * matches and skips the enumeration; otherwise it enumerates -- k = 19 and 20 (1.7 s, 6.7 s) included, which without a cache file
 * are left to the hash-space kernel -- and (re)writes the file.  A stale or foreign file is never used.
 * path NULL or "": no cache.  rk_kmer_cache_state: of the last index build -- 0 none, 1 loaded, 2 enumerated and written, 3
 * enumerated (the file could not be written). */
int rk_set_kmer_cache(rk_ctx* ctx, const char* path);
int rk_kmer_cache_state(const rk_ctx* ctx);

/* Read-depth filter (-M, src/rkmh.cpp:701-704): when set, classify masks hashes whose counter
 * value is below min_kmer_occ (mask_by_frequency, :916) before sketching.  NULL disables.
 * The table is read as it is AT THIS CALL (the fused kernel uses a one-bit-per-slot snapshot of the comparison): set the
 * filter after pass 1 (rk_count_batch*, and after any all-reduce of the table), and again if the table changes later. */
int rk_set_depth_filter(rk_ctx* ctx, rk_counter* counter, int min_kmer_occ);
/* How much of min_num (row field 3) the caller needs while a depth filter is set.  The reference uses num_mins in ONE predicate,
 * `depth_filter = num_mins <= min_matches` (src/rkmh.cpp:938; filter: `read_min_lens[i] <= 0`, :1292), so a caller that compares
 * with n needs min(num_mins, n + 1) and nothing more.  bound < 0 (default): row field 3 is num_mins exactly -- every window is
 * looked up in the depth map.  bound >= 0: row field 3 = min(num_mins, bound); max_id / max_shared / diff are unchanged (the mask
 * changes them only through windows whose hash is a sketch hash: one keep bit per index key, taken when the filter is set), and
 * only the first windows of a read -- until `bound` of them survive -- are looked up by slot.  bound 0 looks up none.
 * May be called before or after rk_set_depth_filter (the snapshot is rebuilt in the other form). */
int rk_set_min_num_bound(rk_ctx* ctx, int bound);
int rk_min_num_bound(const rk_ctx* ctx);

/* Pass 1 of the -M path (src/rkmh.cpp:904-910): hash every read and increment the counter.
 * Passes into ONE counter are ordered by the library (each waits on its stream for the previous one: large batches add their
 * counts with plain stores, not atomics), whatever streams they are given; rk_counter_get / _clear / _add / _copy / _save /
 * _increment and rk_set_depth_filter wait for the passes enqueued so far.  A caller that reads the table through
 * rk_counter_device_ptr() synchronises with the pass's stream itself, as before. */
int rk_count_batch(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, rk_counter* counter);
int rk_count_batch_device(rk_ctx* ctx, const void* d_bases, const void* d_offsets_u32, int64_t nreads,
                          rk_counter* counter, void* hip_stream);

/* The per-read hot loop (src/rkmh.cpp:845-898 / :911-948) for a batch of reads in HOST memory:
 * to_upper -> calc_hashes -> minhashes -> hash_intersection_size vs every reference -> argmax/diff.
 * out4[i*4+0..3] = max_id, max_shared, diff, min_num  (exactly the variables of :874-888).
 * Staged through pinned buffers with hipMemcpyAsync overlapped with the kernels. */
int rk_classify_batch(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int32_t* out4);

/* Same, inputs already resident in HBM (the measured configuration).  d_bases: 4-byte aligned, readable
 * up to the next multiple of 4 past the last base; d_offsets_u32: uint32 [nreads+1] byte offsets into
 * d_bases; d_out4: int32 [nreads*4].  max_read_len: upper bound of the read lengths in the batch (sizes the
 * per-wave LDS image; 0 => determined by a device reduction, which costs one stream sync).
 * hip_stream: the hipStream_t to enqueue on (NULL = HIP's null stream; rk_ctx_stream() = the context's own).
 * Asynchronous: returns after enqueueing.
 * Reads the fused kernel cannot take (longer than 1528 bases, more than 16384 references, or more non-zero hashes than the sketch
 * size) come back with max_id = -2; rk_classify_batch reroutes those itself. */
int rk_classify_batch_device(rk_ctx* ctx, const void* d_bases, const void* d_offsets_u32, int64_t nreads,
                             void* d_out4, uint32_t max_read_len, void* hip_stream);
/* As above, but rows the fused kernel cannot answer (long reads, reads with more windows than the sketch keeps) are
 * answered by the general kernels on the resident bases instead of being flagged; synchronises hip_stream. */
int rk_classify_batch_device_all(rk_ctx* ctx, const void* d_bases, const void* d_offsets_u32, int64_t nreads,
                                 void* d_out4, uint32_t max_read_len, void* hip_stream);

/* hpv16's per-read loop (main_hpv16, src/rkmh.cpp:2656-2719).  The references set with rk_set_reference_sketches are hash
 * LISTS here, not bottom-s sketches: first the argmax_refs HPV type references (all hashes of each, :2546-2547), then the
 * lineage- and sublineage-specific k-mer sets (:2568-2650); sketch_size = the longest list.  Per read: calc_hashes (all -k),
 * mask_by_frequency when a depth filter is set (:2663), sort (:2666); out4 = (max_id, max_shared, diff, non-zero hashes) over
 * the first argmax_refs references with the strict-> first-max rule of :2675-2678; tail_counts[i][nref - argmax_refs] = the
 * intersection size against each remaining reference (what sort_by_similarity ranks, :2688, :2700).  With reference lists
 * of DISTINCT values the intersection is the number of distinct non-zero read hashes present in the list
 * (hash_set_intersection_size; absent from /root/reference, policy U13 of DESIGN.md).  No bottom-s: a read with more hashes
 * than sketch_size is refused (RK_ERR_LIMIT).  Host buffers; runs the general kernels (k_hash_tiles + k_sort_intersect). */
int rk_classify_groups_batch(rk_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, int64_t nreads, int argmax_refs,
                             int32_t* out4, int32_t* tail_counts);

/* ------------------------------------------------------------------------------------------------
 * `call` (main_call, src/rkmh.cpp:1455-1904): k-mer depth map of the reads, sliding-window mean depth along the
 * references (window NOT reset between references, as a single-threaded run of src/rkmh.cpp:1769-1791), and at
 * every position whose depth is below half that mean the 3k SNP and k one-base-deletion k-mers are hashed and
 * looked up; every candidate passing the tests of :1814 / :1853 comes back as one record (rk_free the array).
 * The caller aggregates records into VCF rows (KC/MD/RD/OD, :1821-1829) -- see rkmh_main.cpp. */
typedef struct rk_call_record {
    int32_t ref;        /* reference index */
    int32_t pos;        /* j + alt_pos + 1 (src/rkmh.cpp:1815, :1855) */
    int32_t alt_depth;  /* depth of the rescue k-mer */
    int32_t avg_d;      /* truncated window mean at the position */
    int32_t depth;      /* depth of the original k-mer */
    uint8_t orig, alt;  /* reference base, alternative base ('-' for a deletion) */
    uint8_t kind;       /* 0 SNP, 1 deletion */
    uint8_t pad;
} rk_call_record;
int rk_call(rk_ctx* ctx, const uint8_t* ref_bases, const uint64_t* ref_offsets, int nref,
            const uint8_t* read_bases, const uint64_t* read_offsets, int64_t nreads, int k, int window_len,
            rk_call_record** out, int64_t* nout);

/* Formats one stdout line of stream/classify exactly as src/rkmh.cpp:887-893. Returns bytes written
 * (excluding NUL) or a negative error if cap is too small. */
int rk_format_stream_line(char* dst, size_t cap, const char* ref_name, const char* read_name,
                          int max_shared, int diff, int min_num, int sketch_size, int min_matches, int min_diff);

/* ------------------------------------------------------------------------------------------------
 * FASTA/FASTQ(.gz) front end (parse_fastas, src/rkmh.cpp:238-292; grammar src/kseq.hpp:170-208).
 * Host-side; fills malloc'd concatenated buffers (rk_free each).  names/quals are NUL-separated.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rk_seqset {
    int64_t nseq;
    uint8_t* bases;      /* concatenated, NOT upper-cased here (device does it) */
    uint64_t* offsets;   /* [nseq+1] */
    char* names;         /* name_0 \0 name_1 \0 ... */
    uint64_t* name_offsets; /* [nseq+1] */
    char* quals;         /* concatenated (same offsets as bases) or NULL when any record lacks quals */
} rk_seqset;
int rk_parse_files(const char* const* paths, int npaths, rk_seqset* out);
void rk_seqset_free(rk_seqset* s);
/* rk_seqset_free parks large batch buffers (up to 1.5 GB in all) for the next batch instead of returning them to the system;
 * this releases whatever is parked (a long-lived process that has finished parsing) */
void rk_pool_trim(void);
/* Streaming form (the build's replacement of KSEQ_Reader::get_next_buffer, src/rkmh.cpp:951-959, 2085-2094):
 * up to max_records records / max_bases bases per call (0 = unlimited); out->nseq == 0 at end of input.
 * path "-" reads STDIN.  Uncompressed input read with max_records == 0 or >= 65536 goes through the
 * block-parallel scanner (RKMH_PARSE_THREADS, default 8): a batch is then a whole block of about
 * max_records records, so both limits are approximate there; any input that is not strictly
 * line-structured falls back to the sequential kseq-grammar scanner with identical results. */
typedef struct rk_reader rk_reader;
int rk_reader_open(const char* path, rk_reader** out);
/* the same reader, starting `offset` bytes into the file at a record start (how a file is handed from the device-side FASTQ front
 * end, rk_fastq_slot_*, to this scanner at the first block the device refuses) */
int rk_reader_open_at(const char* path, uint64_t offset, rk_reader** out);
/* the records that START in bytes [lo, hi) of a regular uncompressed FASTQ file (how the ranks of a multi-process run each read
 * their own part of the reads); valid only while rk_reader_strict() stays 1 -- see rk_parse.cpp */
int rk_reader_open_range(const char* path, uint64_t lo, uint64_t hi, rk_reader** out);
int rk_reader_strict(const rk_reader* r);
#define RK_READER_NO_QUALS 1 /* do not keep quality strings (stream/classify never read them) */
void rk_reader_set_options(rk_reader* r, int flags);
int rk_reader_next(rk_reader* r, int64_t max_records, uint64_t max_bases, rk_seqset* out);
void rk_reader_close(rk_reader* r);

/* Synthetic-workload generator of the measured configuration: reads [lo,hi) of the
 * global read set drawn from the reference panel with 1 % substitutions, strand flips and 1-in-1000 'N'.
 * Host-side utility for bench.py and the tests; out holds (hi-lo)*read_len bytes. */
int rk_synth_reads(const uint8_t* ref_bases, const uint64_t* ref_offsets, int nref, uint64_t lo, uint64_t hi,
                   int read_len, uint64_t seed, uint8_t* out, int threads);

/* ------------------------------------------------------------------------------------------------
 * FASTQ text parsed ON THE DEVICE: a block of raw FASTQ text (as it lies in the file, starting at a record start and ending after
 * a record's last newline) is uploaded, split into records, checked, packed and classified by the GPU (rk_fastq.hip) -- the
 * device-side replacement of the host loop  parse_fastas -> kseq_read  (src/rkmh.cpp:238-263; grammar src/kseq.hpp:170-208)
 * followed by the per-read loop (src/rkmh.cpp:845-898), for text that is strictly four lines per record.  For any other text
 * (multi-line sequences, blank lines, carriage returns, '>' / '+' / '@' inside a sequence, a quality string of another length,
 * ...) the call reports status != 0 and leaves the block to the kseq-grammar scanner (rk_reader_* / rk_parse_*): it never guesses.
 * One slot = one block in flight (own stream, page-locked text buffer); several slots of a context may be used from several threads.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rk_fastq_slot rk_fastq_slot;
typedef struct rk_fastq_result {
    int32_t status;            /* 0: the fields below are valid; != 0 (RK_FASTQ_* bits): parse this block on the host instead */
    int64_t nrec;              /* records of the block */
    const int32_t* out4;       /* [nrec][4] rows (max_id, max_shared, diff, min_num) as rk_classify_batch writes them */
    const uint32_t* name_off;  /* name of record i = text[name_off[i] .. name_off[i] + name_len[i])  (header up to the first whitespace) */
    const uint32_t* name_len;
    const uint32_t* seq_off;   /* its sequence = text[seq_off[i] .. + seq_len[i]) */
    const uint32_t* seq_len;
    const uint32_t* qual_off;  /* its quality string = text[qual_off[i] .. + seq_len[i]) */
} rk_fastq_result;             /* the arrays live in the slot and are overwritten by its next call */
#define RK_FASTQ_CR 1          /* a carriage return somewhere in the block */
#define RK_FASTQ_LINES 2       /* lines do not come in fours / no final newline */
#define RK_FASTQ_RECORD 4      /* a record without '@' / '+' line starts, with an empty sequence or a quality string of another length */
#define RK_FASTQ_CHAR 8        /* a sequence byte kseq would not keep, or a quality byte outside 33..127 */
#define RK_FASTQ_CAP 16        /* more records than a block of this size is sized for (records of under 64 bytes on average) */
int rk_fastq_slot_create(rk_ctx* ctx, uint64_t max_bytes, rk_fastq_slot** out);
/* flags: RK_SLOT_DEVICE_TEXT = the block's text never visits the host.  Such a slot has no text buffer of its own (blocks come from
 * rk_fastq_slot_load_bgzf, or from caller memory named with rk_fastq_slot_set_source), may be hundreds of megabytes large at little
 * page-locked memory, and its finish() brings back -- besides the rows -- only the bytes the output needs, packed on the device:
 * the record NAMES (what stream / classify print, src/rkmh.cpp:887-892) or, after rk_fastq_slot_set_filter_output, name, sequence
 * and quality string of the records filter prints (src/rkmh.cpp:1292-1300).  The result's name_off / seq_off / qual_off then index
 * rk_fastq_slot_spans_base() -- which is what the formatters below are to be given as `text` for ANY slot. */
#define RK_SLOT_DEVICE_TEXT 1
int rk_fastq_slot_create2(rk_ctx* ctx, uint64_t max_bytes, int flags, rk_fastq_slot** out);
int rk_fastq_slot_set_filter_output(rk_fastq_slot* slot, int min_matches, int min_diff);
const uint8_t* rk_fastq_slot_spans_base(const rk_fastq_slot* slot);   /* of the last finished block; valid until the slot's next call */
uint8_t* rk_fastq_slot_text(rk_fastq_slot* slot);  /* page-locked buffer of max_bytes: the caller fills it with the block's text (NULL for a device-text slot) */
/* ... or names the text where it lies: the slot's NEXT block (submit / classify / count) is uploaded from `text` -- caller memory,
 * ideally page-locked (a mapping of the input file registered with hipHostRegister: the link reads the page cache, no copy), valid
 * and unchanged until that block's finish / count has returned.  The formatters are given the same pointer. */
int rk_fastq_slot_set_source(rk_fastq_slot* slot, const uint8_t* text);
int rk_host_register_readonly(const void* p, size_t bytes);   /* hipHostRegister of memory that is only read (a file mapping) */
void rk_host_unregister(const void* p);
int rk_fastq_slot_classify(rk_fastq_slot* slot, uint64_t nbytes, rk_fastq_result* res);
/* the same in two halves (classify = submit + finish): submit enqueues the upload and the splitting / checking / packing kernels and
 * returns at once, finish waits, classifies and collects -- a host thread with two slots reads its next block in between */
int rk_fastq_slot_submit(rk_fastq_slot* slot, uint64_t nbytes);
int rk_fastq_slot_finish(rk_fastq_slot* slot, rk_fastq_result* res);
/* pass 1 of -M (src/rkmh.cpp:904-910) on a block of raw text: every window's hash counted into `counter` (a table of the slot's
 * context); *status != 0: the block is not four lines per record and nothing of it was counted */
int rk_fastq_slot_count(rk_fastq_slot* slot, uint64_t nbytes, rk_counter* counter, int32_t* status, int64_t* nrec);

/* The output of stream / classify (line format src/rkmh.cpp:887-892) and of filter (records src/rkmh.cpp:1292-1300, decision
 * src/equiv.hpp:324-353) for one classified block, written from the names, sequences and quality strings where they lie in the
 * block's raw text (rk_fastq_slot_text).  rk_line_parts holds what does not depend on the read ("ref name \t" per reference, the
 * eight possible line tails).  dst must hold *_bound() bytes; the functions return the bytes written or a negative RK_ERR_*. */
typedef struct rk_line_parts rk_line_parts;
int rk_line_parts_create(const char* ref_names, const uint64_t* ref_name_offsets, int64_t nref, int sketch_size, int min_matches,
                         int min_diff, rk_line_parts** out);   /* names NUL-terminated, back to back, as in rk_seqset */
void rk_line_parts_destroy(rk_line_parts* parts);
uint64_t rk_fastq_stream_lines_bound(const rk_line_parts* parts, const rk_fastq_result* res);
int64_t rk_fastq_stream_lines(const rk_line_parts* parts, const rk_fastq_result* res, const uint8_t* text, char* dst, uint64_t cap);
uint64_t rk_fastq_filter_records_bound(const rk_fastq_result* res);
int64_t rk_fastq_filter_records(const rk_fastq_result* res, const uint8_t* text, int min_matches, int min_diff, char* dst, uint64_t cap);

/* ---- references from raw FASTA text (replaces parse_fastas over the -r files, src/rkmh.cpp:238-263 + :816-826, for regular text).
 * The text of the files, concatenated (a '\n' after each file), is uploaded block by block through a slot's page-locked buffer;
 * rk_fasta_load_finish strips header lines and line ends ON THE DEVICE and returns the record table; rk_set_references_fasta
 * sketches the packed bases where they lie.  status != 0 (RK_FASTA_* bits): the text is not plain line-structured FASTA (carriage
 * returns, '>', '+' or '@' inside sequence lines, bases before the first header, bytes outside 33..126) -- parse it on the host. */
typedef struct rk_fasta_load rk_fasta_load;
typedef struct rk_fasta_index {
    int32_t status;
    int64_t nseq;
    const uint64_t* offsets;       /* [nseq + 1] sequence i = packed bases [offsets[i], offsets[i+1]) (on the device) */
    const char* names;             /* NUL-terminated names (header line up to the first whitespace), back to back */
    const uint64_t* name_offsets;  /* [nseq + 1] */
} rk_fasta_index;                  /* the arrays belong to the rk_fasta_load */
#define RK_FASTA_BAD_CHAR 1
#define RK_FASTA_BAD_NAME 2
#define RK_FASTA_BAD_LEAD 4
#define RK_FASTA_EMPTY 8
int rk_fasta_load_create(rk_ctx* ctx, uint64_t text_bytes, rk_fasta_load** out);
int rk_fasta_load_put(rk_fasta_load* load, rk_fastq_slot* via, uint64_t text_offset, uint64_t nbytes);
int rk_fasta_load_finish(rk_fasta_load* load, uint64_t total_bytes, rk_fasta_index* out);
struct rk_gzip;
/* the text of an ordinary gzip file (rk_gzip_open) inflated on the device into the load's text at text_offset; 1: parse on the host instead */
int rk_fasta_load_put_gzip(rk_fasta_load* load, struct rk_gzip* gz, uint64_t text_offset, uint64_t* text_bytes);
/* ... and of members [b0, b1) of a BGZF file (a bgzip'd genome), inflated in the buffers of `via` (RK_SLOT_DEVICE_TEXT); 1: parse on the host */
struct rk_bgzf;
int rk_fasta_load_put_bgzf(rk_fasta_load* load, rk_fastq_slot* via, const struct rk_bgzf* z, int64_t b0, int64_t b1, uint64_t text_offset);
int rk_fasta_load_put_newline(rk_fasta_load* load, uint64_t text_offset);   /* the separator behind such a file */
int rk_fasta_load_get_bases(rk_fasta_load* load, uint8_t* dst);   /* the packed bases, offsets[nseq] bytes, to the host */
int rk_set_references_fasta(rk_ctx* ctx, rk_fasta_load* load, const int* ks, int nks, int sketch_size, int max_samples, uint64_t counter_slots);
void rk_fasta_load_destroy(rk_fasta_load* load);
void rk_fastq_slot_destroy(rk_fastq_slot* slot);
/* Where to cut: the offset of the LAST record start in text[1 .. n) under the four-line rule (a line that begins with '@' whose
 * second line below begins with '+'), or -1 when there is none: text[0 .. offset) then holds whole records only. */
int64_t rk_fastq_cut(const uint8_t* text, uint64_t n);

/* BGZF (bgzip) FASTQ files for the device front end.  The reference opens every input through gzFile (src/rkmh.cpp:238-263): one
 * sequential inflater.  A BGZF file is a chain of independent gzip members (<= 64 KB of text each) whose headers give their
 * lengths, so any number of threads can each inflate the members of their own block of the file.  rk_bgzf_open: RK_ERR_ARG = not a
 * BGZF file (plain gzip, plain text: take another path).  rk_bgzf_plan groups consecutive members into jobs of about target_bytes
 * of text (first[j] .. first[j + 1]); rk_bgzf_fastq_records inflates one job and copies the whole FASTQ records that START in it
 * (four-line rule at both ends, as rk_fastq_cut) to dst, ready for rk_fastq_slot_submit; it returns 1 when the text does not
 * begin with '@'.  Thread-safe on one rk_bgzf. */
typedef struct rk_bgzf rk_bgzf;
int rk_bgzf_open(const char* path, rk_bgzf** out);
void rk_bgzf_close(rk_bgzf* z);
int64_t rk_bgzf_members(const rk_bgzf* z);
uint64_t rk_bgzf_text_bytes(const rk_bgzf* z);
uint64_t rk_bgzf_text_offset(const rk_bgzf* z, int64_t member);
int rk_bgzf_first_byte(const rk_bgzf* z);
int64_t rk_bgzf_plan(const rk_bgzf* z, uint64_t target_bytes, int64_t* first, int64_t cap);
int64_t rk_bgzf_plan_members(const rk_bgzf* z, uint64_t target_bytes, int64_t max_members, int64_t* first, int64_t cap);   /* ... and of at most max_members members each */
int rk_bgzf_fastq_records(const rk_bgzf* z, int64_t b0, int64_t b1, uint8_t* dst, uint64_t cap, uint64_t* nbytes, uint64_t* text_off);
const uint8_t* rk_bgzf_image(const rk_bgzf* z);   /* the mapped file */
/* the member whose text ends with the byte in front of member b0's text (b0 - 1 unless that one is empty; b0 when no text precedes) */
int64_t rk_bgzf_lead_member(const rk_bgzf* z, int64_t b0);
int rk_bgzf_member(const rk_bgzf* z, int64_t member, uint64_t* file_off, uint32_t* total_bytes, uint32_t* header_bytes, uint32_t* text_bytes);
/* The same job inflated ON THE DEVICE (rk_inflate.hip: a lane per member decodes, a wave per member resolves the matches in LDS,
 * another checks the member's CRC-32 -- as gzread does for everything the reference reads, src/rkmh.cpp:238-263): the compressed
 * bytes cross the link instead of the text (by DMA straight from the mapped file when the caller page-locked it:
 * rk_host_register_readonly(rk_bgzf_image(z), rk_bgzf_file_bytes(z))), the records that start in members [b0, b1) land in the slot's
 * device text buffer -- and, for a slot without RK_SLOT_DEVICE_TEXT, a copy of them in rk_fastq_slot_text().  The NEXT
 * rk_fastq_slot_submit / _classify / _count of the slot is given *nbytes and skips its upload.  A job may hold tens of thousands
 * of members (one launch decodes them all: the decode kernel takes the same ~15 ms for 64 members as for 32 768).
 * Returns RK_OK, or 1: take the host route (rk_bgzf_fastq_records) for this job -- it reports damaged members. */
int rk_fastq_slot_load_bgzf(rk_fastq_slot* slot, const rk_bgzf* z, int64_t b0, int64_t b1, uint64_t* nbytes, uint64_t* text_off);

/* ---- ordinary gzip files (ONE deflate stream) inflated on the device: rk_gunzip.hip ------------------------------------------
 * The reference reads every input through gzopen / gzread (/root/reference/src/rkmh.cpp:238-263).  rk_gzip_open maps a gzip file
 * that is not BGZF (RFC 1952 header, deflate payload, trailer); rk_gzip_plan(slot_bytes) cuts its compressed bytes into stretches
 * that inflate to about a slot each (by the ratio the trailer's ISIZE implies), rewinds the stream, and returns how many calls of
 * rk_fastq_slot_load_gzip the file takes.  Each call inflates the next stretch on the device -- block headers found by a kernel, a
 * lane per chunk, the text in front of a chunk resolved in stream order, CRC-32 and ISIZE checked at the end of the stream -- and
 * leaves the whole FASTQ records of it in the slot's device text, for rk_fastq_slot_submit / _count; the bytes behind the last
 * record start wait on the device for the next call.  Calls come in order, all on slots (RK_SLOT_DEVICE_TEXT) of one device.
 * rk_fastq_slot_load_gzip returns RK_OK, or 1: the device route ends here (stored / fixed-code stretches longer than a chunk's
 * scratch, text that outgrows the slot, a second member behind the first, no FASTQ) -- nothing of the text from *text_off on has
 * been delivered, and the caller's sequential reader (rk_reader_open_at(path, *text_off)) continues from there; < 0: damaged data. */
typedef struct rk_gzip rk_gzip;
int rk_gzip_open(const char* path, rk_gzip** out);
void rk_gzip_close(rk_gzip* gz);
const uint8_t* rk_gzip_image(const rk_gzip* gz);        /* the mapped file (rk_host_register_readonly lets the DMA engine read it) */
uint64_t rk_gzip_file_bytes(const rk_gzip* gz);
int rk_gzip_first_byte(const rk_gzip* gz);              /* of the text; -1: not decodable */
uint64_t rk_gzip_text_bytes_hint(const rk_gzip* gz);    /* from ISIZE (modulo 2^32): for sizing only */
int64_t rk_gzip_plan(rk_gzip* gz, uint64_t slot_bytes); /* calls the file takes; rewinds */
int64_t rk_gzip_calls(const rk_gzip* gz);
void rk_gzip_release_device(rk_gzip* gz);              /* frees the device buffers of a file that has been read; rk_gzip_plan before the next pass */
int rk_fastq_slot_load_gzip(rk_fastq_slot* s, rk_gzip* gz, int64_t call, uint64_t* nbytes, uint64_t* text_off);
uint64_t rk_gzip_stretch_bytes(const rk_gzip* gz);      /* compressed bytes per call, after rk_gzip_plan */
int rk_fastq_slot_reserve_gzip(rk_fastq_slot* s, uint64_t comp_bytes); /* optional: the work buffers of the calls, made ahead of the first one */
uint64_t rk_bgzf_file_bytes(const rk_bgzf* z);   /* length of rk_bgzf_image */

/* ------------------------------------------------------------------------------------------------
 * PACKED READS (`rkmh pack`, `stream|filter -F`; the reference parses -F/--pre-reads and does nothing with it, src/rkmh.cpp:659-664).
 * FASTQ text costs ~315 bytes per 150-base read on every hop; a packed file keeps per read 2 bits per base, a 4-byte offset and --
 * for the host only -- its name (and, optionally, its quality string): ~42 bytes per read cross the link, 16 come back.
 * File (little endian; every section 16-byte aligned so a mapping of the file can be uploaded as it lies):
 *   header  rk_packed_header
 *   blocks  per block: offsets u32[nrec + 1] (in BASES, from the block's first base), bases2 u8[ceil(nbases / 4)] (base i in bits
 *           2 (i & 3) of byte i >> 2; A = 0, C = 1, T = 2, G = 3 -- (ASCII >> 1) & 3 --, anything else stored as 0 and listed in),
 *           exceptions {u32 base index, u32 original byte}[nexc] (every byte that is not one of ACGTacgt: N, IUPAC codes ...; lower case
 *           acgt is NOT kept -- mkmh's to_upper folds it before anything looks, src/rkmh.cpp:856), name_offsets u32[nrec + 1], names
 *           (back to back, no terminators), quals u8[nbases] when the file keeps them
 *   directory  rk_packed_block[nblocks] at header.directory_off
 * rk_packed_encode fills one block's device-bound sections from ASCII bases (host code); rk_classify_batch_device_packed /
 * rk_count_batch_device_packed take them from device memory: the bases are expanded to ASCII in HBM (exceptions restored) and go
 * through the same kernels as rk_classify_batch_device_all / rk_count_batch_device -- rows are bit-identical by construction. */
#define RK_PACKED_MAGIC "RKPK1\n\0"
#define RK_PACKED_QUALS 1u
typedef struct rk_packed_header {
    char magic[8];
    uint32_t version, flags;          /* flags: RK_PACKED_QUALS */
    uint64_t nreads, nbases, nblocks;
    uint64_t directory_off;           /* rk_packed_block[nblocks] */
    uint64_t reserved[2];
} rk_packed_header;                   /* 64 bytes */
typedef struct rk_packed_block {
    uint64_t offsets_off, bases_off, exc_off, name_offsets_off, names_off, quals_off;  /* from the start of the file; quals_off 0: none */
    uint32_t nrec, nexc;
    uint64_t nbases, name_bytes;
    uint32_t max_len, pad;
} rk_packed_block;                    /* 80 bytes */
typedef struct rk_packed_exception { uint32_t pos, byte; } rk_packed_exception;
/* bases[0 .. n) -> bases2 (ceil(n / 4) bytes, the last byte's unused bits 0) and the exceptions (at most cap; positions are
 * base_index0 + i); returns the number of exceptions, or a negative error (more than cap).  Host code, thread-safe. */
int64_t rk_packed_encode(const uint8_t* bases, uint64_t n, uint64_t base_index0, uint8_t* bases2, rk_packed_exception* exc, uint64_t cap);
/* the inverse on the host (what filter prints of a passing read): bases [first, first + n) of a block -> ASCII, exceptions restored
 * (exc sorted by pos, as rk_packed_encode leaves them) */
void rk_packed_decode(const uint8_t* bases2, uint64_t first, uint64_t n, const rk_packed_exception* exc, uint32_t nexc, uint8_t* out);
/* d_bases2: ceil(nbases / 4) bytes (4-byte aligned, readable to the next multiple of 4); d_offsets_u32[nreads + 1] in bases;
 * d_exc: nexc pairs (may be NULL when nexc == 0); d_ascii: the caller's scratch, 16 * ceil(nbases / 16) + 64 bytes of device memory,
 * 16-byte aligned -- it holds the batch's ASCII bases on return.  Rows as rk_classify_batch_device_all (no row left flagged);
 * synchronises the stream. */
int rk_classify_batch_device_packed(rk_ctx* ctx, const void* d_bases2, const void* d_offsets_u32, int64_t nreads, uint64_t nbases,
                                    const void* d_exc, uint32_t nexc, void* d_ascii, void* d_out4, uint32_t max_read_len, void* hip_stream);
/* pass 1 of -M on such a batch (src/rkmh.cpp:904-910).  max_read_len: the longest read (the file's directory has it; required).  Asynchronous. */
int rk_count_batch_device_packed(rk_ctx* ctx, const void* d_bases2, const void* d_offsets_u32, int64_t nreads, uint64_t nbases,
                                 const void* d_exc, uint32_t nexc, void* d_ascii, rk_counter* counter, uint32_t max_read_len, void* hip_stream);

/* One block of a packed file in flight (own stream and device arrays): what `stream|filter -F` is made of.  file = the packed file
 * in memory (a mapping; page-locked with rk_host_register_readonly the DMA engine reads the page cache itself).  classify: res->out4 =
 * the rows; name_off / name_len index file + block->names_off (the stream formatter rk_fastq_stream_lines takes that as `text`);
 * seq_off (= qual_off) / seq_len are in BASES from the block's first base -- rk_packed_filter_records decodes what filter prints.
 * The arrays live in the slot and in the mapping until the slot's next call.  count: pass 1 of -M. */
typedef struct rk_packed_slot rk_packed_slot;
int rk_packed_slot_create(rk_ctx* ctx, uint64_t max_reads, uint64_t max_bases, rk_packed_slot** out);
void rk_packed_slot_destroy(rk_packed_slot* slot);
int rk_packed_slot_classify(rk_packed_slot* slot, const rk_packed_block* block, const uint8_t* file, rk_fastq_result* res);
int rk_packed_slot_count(rk_packed_slot* slot, const rk_packed_block* block, const uint8_t* file, rk_counter* counter);
uint64_t rk_packed_filter_records_bound(const rk_fastq_result* res);
int64_t rk_packed_filter_records(const rk_fastq_result* res, const rk_packed_block* block, const uint8_t* file, int min_matches, int min_diff,
                                 char* dst, uint64_t cap);

/* Loads the code objects of the device front end's kernels (with_inflate != 0: and of the device inflater) on `device` ahead of
 * their first launch -- tens of milliseconds a caller can spend on a second thread while its references are sketched.  Optional. */
int rk_warm_up(int device, int with_inflate);

#ifdef __cplusplus
}
#endif
#endif
