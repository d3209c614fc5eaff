// rk_kernels.hpp -- launcher declarations (host side) for the gfx950 kernels in rk_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rk_device.hpp"

namespace rk {

constexpr int MAX_KS = 8;
constexpr int MAX_K = 64;
constexpr int HASH_TILE_WIN = 2048;             // windows per hash tile
constexpr int HASH_TILE_MAXB = HASH_TILE_WIN + MAX_K;
constexpr int FUSED_MAXLEN = 1528;              // longest read taken by the fused classify kernel (prefetch registers)
constexpr int SORT_MAX_P = 16384;               // largest in-LDS sort (128 KiB of u64)

struct KsArr { int32_t n; int32_t k[MAX_KS]; };

// one piece of one sequence to hash with one k
struct TileDesc {
    uint64_t base_off;  // byte offset of the tile's first base in `bases`
    uint64_t out_off;   // index of the tile's first hash in `out`
    uint32_t nbases;    // bases staged (= nwin + k - 1)
    uint32_t nwin;      // windows hashed
    uint32_t k;
    uint32_t pad;
};

// FILTER_KEYMASK: mask_by_frequency through the per-key keep bits of the index (RefIndex::keepkey) instead of the depth table --
// for sequences whose hashes all fit the sketch (no bottom-S selection), where only hashes that ARE index keys matter
enum { FILTER_NONE = 0, FILTER_MASK_MIN = 1, FILTER_RANGE = 2, FILTER_KEYMASK = 3 };

struct SortArgs {
    uint64_t* hashes;          // all hashes (segments per sequence)
    const uint64_t* seg_off;   // [nseq+1]
    const uint32_t* seq_ids;   // sequences handled by this launch
    uint32_t nlist;
    uint32_t P;                // power of two >= longest segment in the list
    int32_t S;
    int32_t write_back;        // write the sorted segment back into `hashes` (minhashes sorts in place)
    uint64_t* sketches;        // nullable: [nseq * S] zero padded
    int32_t* lens;             // nullable
    int32_t* out4;             // nullable: classify results (needs ix)
    const int32_t* counter;    // nullable
    uint64_t slots;
    int32_t filter_mode, fmin, fmax;
    // optional: sort a pre-selected candidate buffer instead of the sequence's own segment (radix-select path);
    // sel_len[0] = number of candidates on the device, filters were already applied
    const uint64_t* sel_hashes;
    const uint32_t* sel_len;
    // optional: sequences with many more hashes than S are not sorted whole; the block first radix-selects the bottom
    // S non-zero hashes of its sequence (12-bit digits, histogram in LDS) straight from global memory and sorts those
    // (P is then next_pow2(S), not the segment length).  Exact for any multiset, including heavy duplicates.
    uint32_t preselect;
    // optional: per-reference counter rows in GLOBAL memory, [gcount_rows][nref] int32 (panels whose counter row does not fit
    // the LDS next to the sort buffer, see sort_intersect_global_rows); the launch then uses at most gcount_rows blocks
    int32_t* gcount;
    uint32_t gcount_rows;
    // optional (hpv16, rkmh.cpp:2656-2719): the argmax/diff only looks at the first argmax_n references (0 = all of them) and
    // the raw shared counts of the others are written to tail_counts[seq id][nref - argmax_n]
    int32_t argmax_n;
    int32_t* tail_counts;
};
// 0 when one block's per-reference counters fit the LDS beside its sort buffer, else the number of global counter rows
// (= blocks) the caller must provide in SortArgs::gcount
uint32_t sort_intersect_global_rows(uint32_t P, int nref);
constexpr int PRESEL_BITS = 12;                 // digit width of the in-block radix select
constexpr int PRESEL_SIDE = 1024;               // threshold-bucket elements resolved by rank sort

// ---- `call` sub-command (rk_call.hip) ----
struct DepthTable {            // exact hash -> occurrence count map in HBM (read_hash_to_depth, rkmh.cpp:1619-1621)
    uint64_t* keys;            // [mask+1], 0 = empty
    uint32_t* counts;          // [mask+1]
    uint32_t* zero_count;      // occurrences of hash 0 (the invalid-k-mer sentinel is a key like any other)
    uint64_t mask;
};
struct CallRecord {            // one candidate that passed the depth tests of rkmh.cpp:1814 / :1853
    int32_t ref, pos, alt_depth, avg_d, depth;
    uint8_t orig, alt, kind, pad;   // kind 0 = SNP, 1 = deletion (alt = '-')
};
hipError_t launch_depth_insert(const uint64_t* h, uint64_t n, const DepthTable& t, hipStream_t st);
hipError_t launch_depth_lookup(const uint64_t* h, uint64_t n, const DepthTable& t, int32_t* depth, hipStream_t st);
hipError_t launch_exclusive_scan(const int32_t* in, uint64_t n, int64_t* out, int64_t* scratch, hipStream_t st);
hipError_t launch_call_enumerate(const uint8_t* ref_upper, const uint64_t* ref_off, const uint64_t* win_off, int nref,
                                 uint64_t nwin_total, const int32_t* depth, const int64_t* prefix, int k, int window_len,
                                 const DepthTable& t, const DevPolicy& pol, CallRecord* out, uint32_t* out_count, uint32_t out_cap,
                                 hipStream_t st);

hipError_t launch_to_upper(uint8_t* d, uint64_t n, hipStream_t st);
hipError_t launch_hash_tiles(const uint8_t* bases, const TileDesc* tiles, uint32_t ntiles, uint64_t* out,
                             int32_t* counter, uint64_t slots, const DevPolicy& pol, hipStream_t st);
hipError_t launch_sort_intersect(const SortArgs& a, const RefIndex* ix, const DevPolicy& pol, hipStream_t st);
// bottom-S candidates of one LONG sequence (more hashes than the in-LDS sorter holds): exact radix select of the
// S-th smallest kept hash (13-bit digits), then compaction into sel_out[<= S]; sel_state is 16 dwords of scratch,
// hist 8192 dwords.  sel_state[8] receives the number of candidates.
hipError_t launch_select_bottom(const uint64_t* hashes, uint64_t n, int S, const int32_t* counter, uint64_t slots,
                                int filter_mode, int fmin, int fmax, const DevPolicy& pol, uint32_t* sel_state,
                                uint32_t* hist, uint64_t* sel_out, hipStream_t st);
// ---- FASTQ text indexed on the device (rk_fastq.hip) ----
enum { FQ_BAD_CR = 1, FQ_BAD_LINES = 2, FQ_BAD_RECORD = 4, FQ_BAD_CHAR = 8, FQ_BAD_CAP = 16 }; // status bits: any of them = parse this block on the host
struct FqDev {
    uint32_t* chunk_cnt;   // [chunks + 1] newlines per 4 KB chunk
    uint32_t* chunk_base;  // [chunks + 1] their exclusive scan; [chunks] = lines of the block
    uint32_t* nl;          // [line_cap] newline positions
    uint32_t line_cap, rec_cap;
    uint32_t *seq_off, *seq_len, *qual_off, *name_off, *name_len; // [rec_cap + 1]
    uint32_t* out_off;     // [rec_cap + 1] offsets of the packed batch
    uint8_t* bases;        // packed batch (block bytes + slack)
    uint32_t* info;        // [4] status bits, records, longest sequence, bytes packed by launch_fastq_pack
    void* scan_tmp;
    size_t scan_tmp_bytes;
    // launch_fastq_pack: the spans of the text the HOST still needs, back to back in `pack` (pack_cap bytes + 16 of slack)
    uint32_t* pack_len;    // [rec_cap + 1] bytes record r contributes
    uint32_t* pack_off;    // [rec_cap + 1] their exclusive scan: where record r's spans begin in pack
    uint8_t* pack;
    uint64_t pack_cap;
};
size_t fq_scan_temp_bytes(uint32_t n);
hipError_t launch_fastq_index(const FqDev& d, const uint8_t* raw, uint64_t nbytes, hipStream_t st);
// After launch_fastq_index (and, for the filter form, after the rows are final): what the output formatters read of each record,
// copied out of the raw text into d.pack so that only those bytes travel to the host.  names_only: every record's name (stream /
// classify lines, src/rkmh.cpp:887-892).  Otherwise name, sequence and quality string of the records filter prints
// (rk_filter_keeps on rows out4: src/rkmh.cpp:1292-1300).  Record r's spans begin at pack_off[r]: name, then sequence, then quality.
// info[3] = bytes packed; FQ_BAD_CAP in info[0] when they exceed pack_cap (nothing is trusted then).
hipError_t launch_fastq_pack(const FqDev& d, const uint8_t* raw, bool names_only, const int32_t* out4, int min_matches, int min_diff, hipStream_t st);
// ---- reference FASTA text stripped on the device (rk_fasta.hip) ----
enum { FA_BAD_CHAR = 1, FA_BAD_NAME = 2, FA_BAD_LEAD = 4, FA_BAD_EMPTY = 8 }; // status bits: any of them = parse the files on the host
struct FaDev {
    uint32_t *chunk_map, *chunk_pre;    // [chunks] state map of each 4 KB chunk; map of everything before it
    uint64_t *chunk_kept, *chunk_hdrs;  // [chunks + 1] bases kept / headers begun per chunk
    uint64_t *kept_base, *hdr_base;     // [chunks + 1] their exclusive sums; [chunks] = the totals
    uint8_t* bases;                     // packed bases
    uint64_t *hdr_pos, *rec_off;        // [records] text position of each '>' / offset of the record's first base
    uint64_t *name_len1, *name_off;     // [records + 1] name length + 1 / its exclusive sum
    uint8_t* names;                     // NUL-terminated names, back to back
    uint32_t* info;                     // [4] status bits
    void* scan_tmp;
    size_t scan_tmp_bytes;
};
size_t fa_scan_temp_bytes(uint64_t n);
uint64_t fa_chunks(uint64_t nbytes);
hipError_t launch_fasta_count(const FaDev& d, const uint8_t* raw, uint64_t nbytes, hipStream_t st);
hipError_t launch_fasta_compact(const FaDev& d, const uint8_t* raw, uint64_t nbytes, uint64_t nrec, hipStream_t st);
hipError_t launch_fasta_names(const FaDev& d, const uint8_t* raw, uint64_t nrec, hipStream_t st);
// ---- DEFLATE on the device for BGZF members (rk_inflate.hip): pass 1 one lane per member, pass 2 one wave per member ----
// deflate payload in the compressed buffer (which has 64 readable bytes beyond the last member); text in the output buffer;
// match_off = first dword of the member's part of `scratch` (inflate_scratch_dwords(out_len) dwords: its entries from the front, its literals from the back)
struct InflateMember { uint32_t in_off, in_len, out_off, out_len, match_off, pad; };
uint32_t inflate_scratch_dwords(uint32_t out_len);
// status[0 .. nmem) = 0 / why the member could not be inflated, status[nmem .. 2 nmem) = its entries
hipError_t launch_inflate_members(const uint8_t* comp, uint32_t comp_bytes, const InflateMember* mem, uint32_t nmem, uint8_t* text, uint32_t* scratch, uint32_t* status,
                                  hipStream_t st);
// ---- an ordinary gzip stream on the device (kernels in rk_inflate.hip, driven by rk_gunzip.hip) ----
// A chunk of the stream: pass 1 decodes it from start_bit (a block header; bit offsets count from the compressed buffer's first byte) to
// the first block boundary at or after stop_bit, into its part of the scratch buffer (scratch_dw dwords at dword scratch_off: entries
// from the front, literals from the back), and reports where it ended and what it made; the host then gives the chunks that form
// the stream their places in the text (out_off) for pass 2.
struct GzChunk {
    uint32_t start_bit, stop_bit, scratch_off, scratch_dw;
    uint32_t flags;      // 1: the stream's first chunk (a match may not reach in front of it)
    uint32_t status;     // out: 0, or why the lane gave up
    uint32_t end_bit;    // out: the block boundary where it stopped
    uint32_t final_seen; // out: 1 = it ended with the stream's last block
    uint32_t out_len, nent, nlit; // out: bytes of text, entries, literals
    uint32_t out_off;    // host: its text's place (pass 2)
};
hipError_t launch_gz_find_starts(const uint8_t* comp, uint32_t nbits, const uint32_t* from, const uint32_t* to, uint32_t n, uint32_t* found, hipStream_t st);
hipError_t launch_gz_lanes(const uint8_t* comp, uint32_t comp_bytes, GzChunk* chunks, uint32_t n, uint32_t* scratch, hipStream_t st);
// pass 2 of n chunks in stream order, `group` consecutive chunks placed together as one unit (units[u]: out_off / out_len of unit u,
// nunits = ceil(n / group)): three planes (plane_stride apart), the windows in front of the units (rings: (nunits + 1) x 32 KB,
// heads: nunits + 1 -- [0] given), the text (text[out_off ..) of every chunk), and the CRC-32 of every 64 KB segment of text[0, text_bytes)
hipError_t launch_gz_place(const GzChunk* chunks, uint32_t n, uint32_t group, const GzChunk* units, uint32_t nunits, uint8_t* planes, size_t plane_stride, const uint32_t* scratch,
                           uint8_t* rings, uint32_t* heads, uint8_t* text, uint32_t text_bytes, uint32_t* crc, hipStream_t st);
// cuts[which] = first FASTQ record start (four-line rule) at or after `from`; n: none and the text ends here; 0xFFFFFFFF: not decidable from this text
hipError_t launch_fastq_first_start(const uint8_t* text, uint32_t n, uint32_t from, uint32_t window, bool at_eof, uint32_t* cuts, int which, hipStream_t st);
// whole-array ascending sort of u64 keys in place (rk_sort.hip: rocPRIM radix sort); tmp holds sort_u64_temp_bytes(n) bytes
hipError_t sort_u64_temp_bytes(uint64_t n, size_t* bytes);
hipError_t launch_sort_u64(uint64_t* keys, uint64_t n, void* tmp, size_t tmp_bytes, hipStream_t st);
// bits[s / 32] bit (s % 32) = the count of slot s passes mask_by_frequency's threshold (one streaming pass over the table)
hipError_t launch_keep_bits(const int32_t* counter, uint64_t slots, int min_occ, const DevPolicy& pol, uint32_t* bits, hipStream_t st);
// -M with a bounded min_num: keep bit per index key, the masked copy of the exact k-mer map, min(min_num, bound) per read
hipError_t launch_keep_keys(const RefIndex& ix, uint32_t nkeys, const int32_t* counter, uint64_t slots, const uint32_t* key_sid, int min_occ,
                            const DevPolicy& pol, uint32_t* bits, hipStream_t st);
hipError_t launch_kv_mask(const uint4* kv, uint32_t nkeys, const uint32_t* keepkey, uint4* kvm, hipStream_t st);
hipError_t launch_km1_mask(const uint2* cells, uint32_t n, const uint32_t* keepkey, uint32_t* km1m, uint32_t vmask, hipStream_t st);
hipError_t launch_min_num_probe(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S, int bound,
                                const uint32_t* keepbits, uint64_t slots, const DevPolicy& pol, int32_t* out4, hipStream_t st);
// hashes all 4^k k-mers; stats[0] = k-mers found (one per strand pair), the first list_cap of them in list[] as (packed canonical
// k-mer, key id or IDX_NOT_FOUND for a zero hash)
hipError_t launch_enum_kmers(const RefIndex& ix, const DevPolicy& pol, int k, uint32_t* stats, uint2* list, uint32_t list_cap,
                             hipStream_t st);
hipError_t launch_intersect_pair(const uint64_t* a, int na, const uint64_t* b, int nb, int* out, hipStream_t st);
hipError_t launch_intersect_pair_emit(const uint64_t* a, int na, const uint64_t* b, int nb, int cap, uint64_t* out, int* n_out, hipStream_t st);
hipError_t launch_fill_reroute(int32_t* out4, uint32_t nreads, hipStream_t st);
hipError_t launch_scatter_rows(const int32_t* rows, const uint32_t* ids, uint32_t m, int32_t* out4, hipStream_t st);
// The compact depth map (rk_counter_create_compact): only the slots some index key maps to are counted, entry e of the map
// counts the e-th smallest of them.  pre: one bit per hashed slot (an L2-resident filter: a window whose slot fails it is not
// counted at all), tab: open-addressing (slot, entry) pairs, empty = slot CS_EMPTY.
constexpr uint32_t CS_EMPTY = 0xFFFFFFFFu;
struct CompactSlots {
    const uint32_t* pre;
    const uint2* tab;
    uint32_t pre_shift;  // bit index = (slot * 0x9E3779B1) >> pre_shift
    uint32_t tab_shift;  // first probe = (slot * 0x85EBCA6B) >> tab_shift
    uint32_t tab_mask;
};
// mode 0: classify (out4 written); mode 1: count only (counter incremented)
// wave-per-tile fused kernel (rk_classify.hip); expect_hits sizes the per-read hit multiset
bool classify_tile_supported(int nref, int maxlen);
constexpr int KPRE_MIN_K = 8; // the k-mer-space kernel (rk_kmer.hip) exists for a single k in [KPRE_MIN_K, 16]
hipError_t launch_classify_tile(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S,
                                const RefIndex& ix, int32_t* counter, uint64_t slots, int min_occ, int mode,
                                int32_t* out4, const DevPolicy& pol, int maxlen, int expect_hits, hipStream_t st,
                                uint32_t slot_stride = 0, int nmin_cap = 0x7fffffff, const CompactSlots* compact = nullptr);
// the k-mer-space kernel (rk_kmer.hip): plain classification with k-mer sizes from KPRE_MIN_K to 16 whose exact k-mer maps and group
// filters were built (KmerSets: one per size; a single size runs the compile-time-k kernels, several the run-time-k one)
bool classify_kmer_supported(int nref, int maxlen, int k);
hipError_t launch_classify_kmer(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KmerSets& ksets, int S, const RefIndex& ix,
                                int32_t* out4, const DevPolicy& pol, int maxlen, int expect_hits, hipStream_t st, int nmin_cap = 0x7fffffff);
hipError_t launch_max_len(const uint32_t* offs, uint32_t nreads, uint32_t* d_max, hipStream_t st);
hipError_t launch_mask_by_frequency(uint64_t* h, uint64_t n, const int32_t* counter, uint64_t slots, int min_occ,
                                    const DevPolicy& pol, hipStream_t st);
// increments counter[h % slots] once per DISTINCT hash of one sample (filter's hash_sequences, rkmh.cpp:348-355):
// exact set semantics through a scratch open-addressing table of tsize (power of two, >= 2n) u64 slots, zeroed here
hipError_t launch_count_distinct(const uint64_t* hashes, uint64_t n, uint64_t* table, uint64_t tsize, int32_t* counter,
                                 uint64_t slots, hipStream_t st);
hipError_t launch_counter_inc(int32_t* counter, uint64_t slots, uint64_t key, hipStream_t st);
// ---- the count pass without global atomics (rk_count.hip) ----
constexpr int CB_THREADS = 1024;
#ifndef RK_CB_CHUNK
#define RK_CB_CHUNK 16384
#endif
constexpr int CB_CHUNK = RK_CB_CHUNK;     // slots counting-sorted per step (64 KB of LDS); a multiple of 4096
constexpr int CB_SUB_LG = 15;             // log2(slots per sub-range) = counters held in LDS (128 KB)
constexpr int CB_MAX_BINS = 1024;
constexpr uint32_t CB_NONE = 0xFFFFFFFFu; // "no window at this position"
struct CountPlan {
    uint64_t slots, n, span; // table size; entries of the slot array (multiple of 4); entries per span (multiple of CB_CHUNK)
    uint32_t nsub, R, nb, magicR, G; // sub-ranges of 2^15 slots; sub-ranges per bin; bins; ceil(2^32 / R); spans
};
struct CountScratch { uint32_t *flat, *binned, *hist, *off, *total, *bin_start; };
// false: this table / batch stays on the atomic form (>= 2^32 - 1 slots, >= 2^31 entries)
bool count_plan(uint64_t slots, uint64_t n_entries, CountPlan* out);
size_t count_plan_scratch_bytes(const CountPlan& pl);
CountScratch count_plan_carve(const CountPlan& pl, void* ws);
hipError_t launch_count_prepare(const CountPlan& pl, const CountScratch& s, hipStream_t st); // before the slot-emitting launch_classify_tile
hipError_t launch_count_bins(const CountPlan& pl, const CountScratch& s, int32_t* counter, hipStream_t st); // after it
hipError_t launch_counter_add(int32_t* dst, const int32_t* src, uint64_t n, hipStream_t st); // dst[i] += src[i]; both 16-byte aligned

} // namespace rk
