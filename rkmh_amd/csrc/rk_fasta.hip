// rk_fasta.hip -- reference FASTA text to packed bases on the device (gfx950).  The whole text of the -r files is uploaded as it
// lies on disk; the GPU strips the header lines and the line ends and leaves the concatenated bases (what rk_set_references takes)
// in HBM, together with every record's offset and its name for the host.  Replaces, for regular text, the host-side parse_fastas
// loop over the references (/root/reference/src/rkmh.cpp:238-263, record grammar /root/reference/src/kseq.hpp:170-208) -- at
// BASELINE config 4's size (a 3.1 Gb genome) the host parser was the longest stage of the run.
//
// As in rk_fastq.hip the device only accepts text on which kseq's sequential grammar provably coincides with the line-oriented
// one: a line that begins with '>' is a header (name = up to the first whitespace), every other line holds only keeper bytes
// (33..126 without '>', '+', '@': a line beginning with '+' or '@' would switch kseq to its quality state, kseq.hpp:192-208), no
// carriage returns, nothing but blank lines before the first header.  Anything else sets a status bit and the host parser takes
// the files (fail closed).
//
// One pass cannot tell whether a byte belongs to a header: that depends on the last line start before it, which may lie any
// distance back (a chromosome on ONE line is legal).  Every 16-byte piece, and from them every 4 KB chunk, is therefore summarised
// as a MAP from the state at its first byte to the state after its last byte, over three states (inside a sequence line, inside a
// header line, at a line start); maps compose associatively, so the state at every chunk's first byte is an exclusive scan of
// the chunk maps (rocPRIM, custom operator).  Kernels, all HBM-streaming:
//   k_fa_maps      chunk maps
//   (scan)         state at each chunk start
//   k_fa_count     bases kept and headers begun per chunk; keeper check
//   (scans)        where each chunk's bases go, which record numbers its headers get
//   k_fa_compact   bases -> packed array; per record: text position of its header, offset of its first base
//   k_fa_name_len / (scan) / k_fa_name_copy   names, NUL-terminated, into one blob for the host
#include "rk_kernels.hpp"

#include <hipcub/hipcub.hpp>

namespace rk {

namespace {

constexpr int FA_CHUNK = 4096; // 256 threads x 16 bytes
enum : uint32_t { ST_SEQ = 0, ST_HDR = 1, ST_FRESH = 2 };
constexpr uint32_t MAP_ID = ST_SEQ | (ST_HDR << 2) | (ST_FRESH << 4);

__host__ __device__ __forceinline__ uint32_t map_const(uint32_t s) { return s | (s << 2) | (s << 4); }
__host__ __device__ __forceinline__ uint32_t map_apply(uint32_t m, uint32_t s) { return (m >> (2u * s)) & 3u; }
// the map of "a, then b"
__host__ __device__ __forceinline__ uint32_t map_then(uint32_t a, uint32_t b) {
    return map_apply(b, a & 3u) | (map_apply(b, (a >> 2) & 3u) << 2) | (map_apply(b, (a >> 4) & 3u) << 4);
}
struct MapThen {
    __host__ __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return map_then(a, b); }
};

struct Piece16 {
    uint8_t b[16];
    int nv; // bytes of the piece inside the text
};
__device__ __forceinline__ Piece16 load_piece(const uint8_t* raw, uint64_t nbytes, uint64_t chunk, int tid) {
    Piece16 p;
    const uint64_t pos = chunk * FA_CHUNK + (uint64_t)tid * 16;
    const uint4 v = reinterpret_cast<const uint4*>(raw)[pos >> 4]; // the buffer is padded to whole chunks
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) p.b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
    p.nv = pos >= nbytes ? 0 : (nbytes - pos >= 16 ? 16 : (int)(nbytes - pos));
    return p;
}
__device__ __forceinline__ uint32_t piece_map(const Piece16& p) {
    if (p.nv == 0) return MAP_ID;
    int last = -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < p.nv && p.b[i] == '\n') last = i;
    if (last < 0) return ST_SEQ | (ST_HDR << 2) | ((p.b[0] == '>' ? ST_HDR : ST_SEQ) << 4);
    if (last == p.nv - 1) return map_const(ST_FRESH);
    uint8_t nx = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) if (i == last + 1) nx = p.b[i];
    return map_const(nx == '>' ? ST_HDR : ST_SEQ);
}

// exclusive scan of the 256 threads' maps (and the block's total) -- shuffles inside a wave, four wave totals through LDS
__device__ __forceinline__ uint32_t block_map_scan(uint32_t m, uint32_t* wtot, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl = map_then(o, incl);
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t pre = MAP_ID;
    for (int i = 0; i < wv; ++i) pre = map_then(pre, wtot[i]);
    total = map_then(map_then(map_then(wtot[0], wtot[1]), wtot[2]), wtot[3]);
    uint32_t ex = (uint32_t)__shfl_up((int)incl, 1);
    if (lane == 0) ex = MAP_ID;
    __syncthreads(); // wtot may be reused by the caller
    return map_then(pre, ex);
}
// exclusive scan of 256 counts
__device__ __forceinline__ uint32_t block_sum_scan(uint32_t v, uint32_t* wtot, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (int i = 0; i < wv; ++i) pre += wtot[i];
    total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    return pre + incl - v;
}

__global__ __launch_bounds__(256) void k_fa_maps(const uint8_t* __restrict__ raw, uint64_t nbytes, uint32_t* __restrict__ chunk_map) {
    __shared__ uint32_t wtot[4];
    const Piece16 p = load_piece(raw, nbytes, blockIdx.x, threadIdx.x);
    uint32_t total;
    block_map_scan(piece_map(p), wtot, total);
    if (threadIdx.x == 0) chunk_map[blockIdx.x] = total;
}

// the walk over one piece from state s: bases kept, headers begun, keeper violations
template <typename KeepFn, typename HdrFn>
__device__ __forceinline__ uint32_t walk_piece(const Piece16& p, uint32_t s, KeepFn keep, HdrFn hdr) {
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i >= p.nv) break;
        const uint32_t c = p.b[i];
        if (c == '\n') { s = ST_FRESH; continue; }
        if (s == ST_FRESH) {
            if (c == '>') { s = ST_HDR; hdr(i); continue; }
            s = ST_SEQ;
        }
        if (s == ST_SEQ) {
            if ((c - 33u) > 93u || c == '>' || c == '+' || c == '@') bad = FA_BAD_CHAR;
            keep(i, (uint8_t)c);
        }
    }
    return bad;
}

__global__ __launch_bounds__(256) void k_fa_count(const uint8_t* __restrict__ raw, uint64_t nbytes, const uint32_t* __restrict__ chunk_pre,
                                                  uint64_t* __restrict__ chunk_kept, uint64_t* __restrict__ chunk_hdrs, uint32_t* __restrict__ info) {
    __shared__ uint32_t wtot[4];
    const Piece16 p = load_piece(raw, nbytes, blockIdx.x, threadIdx.x);
    uint32_t total;
    const uint32_t ex = block_map_scan(piece_map(p), wtot, total);
    const uint32_t s = map_apply(ex, map_apply(chunk_pre[blockIdx.x], ST_FRESH));
    uint32_t kept = 0, hdrs = 0;
    const uint32_t bad = walk_piece(p, s, [&](int, uint8_t) { ++kept; }, [&](int) { ++hdrs; });
    if (bad) atomicOr(&info[0], bad);
    uint32_t tk, th;
    block_sum_scan(kept, wtot, tk);
    block_sum_scan(hdrs, wtot, th);
    if (threadIdx.x == 0) { chunk_kept[blockIdx.x] = tk; chunk_hdrs[blockIdx.x] = th; }
}

__global__ __launch_bounds__(256) void k_fa_compact(const uint8_t* __restrict__ raw, uint64_t nbytes, const uint32_t* __restrict__ chunk_pre,
                                                    const uint64_t* __restrict__ kept_base, const uint64_t* __restrict__ hdr_base,
                                                    uint8_t* __restrict__ bases, uint64_t* __restrict__ hdr_pos, uint64_t* __restrict__ rec_off,
                                                    uint64_t rec_cap) {
    __shared__ uint32_t wtot[4];
    const Piece16 p = load_piece(raw, nbytes, blockIdx.x, threadIdx.x);
    uint32_t total;
    const uint32_t ex = block_map_scan(piece_map(p), wtot, total);
    const uint32_t s = map_apply(ex, map_apply(chunk_pre[blockIdx.x], ST_FRESH));
    uint32_t kept = 0, hdrs = 0;
    walk_piece(p, s, [&](int, uint8_t) { ++kept; }, [&](int) { ++hdrs; });
    uint32_t tk, th;
    const uint32_t k0 = block_sum_scan(kept, wtot, tk);
    const uint32_t h0 = block_sum_scan(hdrs, wtot, th);
    uint64_t o = kept_base[blockIdx.x] + k0;
    uint64_t r = hdr_base[blockIdx.x] + h0;
    const uint64_t pos = (uint64_t)blockIdx.x * FA_CHUNK + (uint64_t)threadIdx.x * 16;
    walk_piece(p, s, [&](int, uint8_t c) { bases[o++] = c; },
               [&](int i) { if (r < rec_cap) { hdr_pos[r] = pos + (uint64_t)i; rec_off[r] = o; } ++r; });
}

__device__ __forceinline__ bool fa_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }

__global__ __launch_bounds__(256) void k_fa_name_len(const uint8_t* __restrict__ raw, uint64_t nbytes, const uint64_t* __restrict__ hdr_pos,
                                                     uint64_t nrec, uint64_t* __restrict__ name_len1, uint32_t* __restrict__ info) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    uint64_t q = hdr_pos[r] + 1;
    const uint64_t q0 = q;
    while (q < nbytes && !fa_space(raw[q]) && q - q0 < (1u << 16)) ++q;
    if (q - q0 >= (1u << 16)) atomicOr(&info[0], (uint32_t)FA_BAD_NAME);
    name_len1[r] = q - q0 + 1; // with its NUL
}
__global__ __launch_bounds__(256) void k_fa_name_copy(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ hdr_pos, uint64_t nrec,
                                                      const uint64_t* __restrict__ name_off, uint8_t* __restrict__ names) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    const uint64_t a = name_off[r], n = name_off[r + 1] - a - 1;
    const uint8_t* src = raw + hdr_pos[r] + 1;
    for (uint64_t i = 0; i < n; ++i) names[a + i] = src[i];
    names[a + n] = 0;
}

} // namespace

size_t fa_scan_temp_bytes(uint64_t n) {
    size_t a = 0, b = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveScan(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, MapThen(), MAP_ID, (int)n, nullptr);
    (void)e;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, nullptr);
    (void)e;
    return (a > b ? a : b) + 256;
}

uint64_t fa_chunks(uint64_t nbytes) { return (nbytes + FA_CHUNK - 1) / FA_CHUNK; }

// stage 1: maps, states, counts, the two scans.  Afterwards kept_base[chunks] = bases in all, hdr_base[chunks] = records.
hipError_t launch_fasta_count(const FaDev& d, const uint8_t* raw, uint64_t nbytes, hipStream_t st) {
    const uint64_t chunks = fa_chunks(nbytes);
    if (chunks == 0 || chunks >= 0x7fffffffull) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d.info, 0, 16, st);
    if (e != hipSuccess) return e;
    k_fa_maps<<<dim3((uint32_t)chunks), dim3(256), 0, st>>>(raw, nbytes, d.chunk_map);
    size_t tb = d.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveScan(d.scan_tmp, tb, d.chunk_map, d.chunk_pre, MapThen(), MAP_ID, (int)chunks, st);
    if (e != hipSuccess) return e;
    // (one element more than there are chunks, zero: the exclusive sums then end with the totals)
    e = hipMemsetAsync(d.chunk_kept + chunks, 0, 8, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d.chunk_hdrs + chunks, 0, 8, st);
    if (e != hipSuccess) return e;
    k_fa_count<<<dim3((uint32_t)chunks), dim3(256), 0, st>>>(raw, nbytes, d.chunk_pre, d.chunk_kept, d.chunk_hdrs, d.info);
    tb = d.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.chunk_kept, d.kept_base, (int)chunks + 1, st);
    if (e != hipSuccess) return e;
    tb = d.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.chunk_hdrs, d.hdr_base, (int)chunks + 1, st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// stage 2 (the caller has sized bases / hdr_pos / rec_off / name_len1 from stage 1's totals): bases packed, record table, name lengths
hipError_t launch_fasta_compact(const FaDev& d, const uint8_t* raw, uint64_t nbytes, uint64_t nrec, hipStream_t st) {
    const uint64_t chunks = fa_chunks(nbytes);
    k_fa_compact<<<dim3((uint32_t)chunks), dim3(256), 0, st>>>(raw, nbytes, d.chunk_pre, d.kept_base, d.hdr_base, d.bases, d.hdr_pos, d.rec_off, nrec);
    if (nrec) {
        k_fa_name_len<<<dim3((uint32_t)((nrec + 255) / 256)), dim3(256), 0, st>>>(raw, nbytes, d.hdr_pos, nrec, d.name_len1, d.info);
        hipError_t e = hipMemsetAsync(d.name_len1 + nrec, 0, 8, st);
        if (e != hipSuccess) return e;
        size_t tb = d.scan_tmp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.name_len1, d.name_off, (int)nrec + 1, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// stage 3 (names sized from name_off[nrec])
hipError_t launch_fasta_names(const FaDev& d, const uint8_t* raw, uint64_t nrec, hipStream_t st) {
    if (nrec) k_fa_name_copy<<<dim3((uint32_t)((nrec + 255) / 256)), dim3(256), 0, st>>>(raw, d.hdr_pos, nrec, d.name_off, d.names);
    return hipGetLastError();
}

} // namespace rk
