// rk_fastq.hip -- the FASTQ front end on the device (gfx950): a block of RAW FASTQ text, uploaded as it lies in the file, becomes
// the packed batch the classify kernels read (concatenated bases + offsets) plus, for the host, where each record's name lies in
// the raw text.  Replaces, for input that is strictly four lines per record, the host-side parse_fastas / kseq_read loop
// (/root/reference/src/rkmh.cpp:238-263; record grammar /root/reference/src/kseq.hpp:170-208) that round 3 measured as the
// stage the GPU waits for (33 M reads/s on 16 cores against a 3 G reads/s kernel).
//
// kseq's grammar is sequential (a quality string is read by COUNT, '@' may be data), so -- exactly like the block-parallel host
// scanner in rk_parse.cpp -- the device only accepts text on which that grammar provably coincides with the line-oriented one:
//   * the block starts at a record start and its lines come in fours: '@' header, sequence, '+' line, quality;
//   * the sequence line holds only keeper bytes (33..126 without '>', '+', '@': what kseq appends to seq.s, kseq.hpp:183-191);
//   * the quality line is exactly as long and holds bytes 33..127;
//   * no carriage returns and no empty records (a line structure kseq would read differently is never guessed at).
// ANY deviation sets a status bit and the caller parses the block with the sequential scanner instead (fail closed).
//
// Kernels (all HBM-streaming, no atomics on the data path):
//   k_fq_count     newlines per 4 KB chunk (ballot-free: byte masks of 16-byte loads, popcounts, a block reduction)
//   (exclusive scan of the chunk counts: rocPRIM)
//   k_fq_positions position of every newline, in order (intra-block scan of per-thread counts)
//   k_fq_records   one thread per record: the four line starts, the structural checks, name span, sequence span, longest read
//   (exclusive scan of the sequence lengths = the batch's offsets: rocPRIM)
//   k_fq_gather    one wave per record (grid-stride): sequence bytes -> packed batch, keeper / quality range checks
#include "rk_kernels.hpp"
#include "rk_filter_rule.hpp"

#include <hipcub/hipcub.hpp>

namespace rk {

namespace {

constexpr int FQ_CHUNK = 4096; // bytes per workgroup of the newline kernels: 256 threads x 16 bytes

// bit i set <=> byte i of the 16 bytes is '\n'; cr collects whether any byte is '\r'
__device__ __forceinline__ uint32_t nl_mask16(const uint4& v, uint32_t& cr) {
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t x = w[d] ^ 0x0A0A0A0Au; // zero bytes where '\n'
        const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); // 0x80 in every zero byte
        m |= (((z >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * d); // gathers the four flag bits (bits 0, 8, 16, 24) into a nibble
        const uint32_t y = w[d] ^ 0x0D0D0D0Du;
        cr |= ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
    }
    return m;
}
// the 16 bytes at raw + 16 i, bytes at or past nbytes read as 'A' (neither '\n' nor '\r'); raw is 16-byte aligned and padded
__device__ __forceinline__ uint4 load16(const uint8_t* raw, uint64_t nbytes, uint64_t i) {
    uint4 v = reinterpret_cast<const uint4*>(raw)[i];
    const uint64_t b = i * 16;
    if (b + 16 > nbytes) {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (b + (uint64_t)(4 * d + j) >= nbytes) w[d] = (w[d] & ~(0xFFu << (8 * j))) | (0x41u << (8 * j));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_fq_count(const uint8_t* __restrict__ raw, uint64_t nbytes, uint32_t* __restrict__ chunk_cnt,
                                                  uint32_t* __restrict__ status) {
    __shared__ uint32_t part[4];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cr = 0, n = 0;
    if (i * 16 < nbytes) n = (uint32_t)__popc(nl_mask16(load16(raw, nbytes, i), cr));
    if (cr) atomicOr(status, FQ_BAD_CR);
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// chunk_base = exclusive scan of chunk_cnt ([nchunks] = total); lines beyond cap are not written (the caller sees total > cap)
__global__ __launch_bounds__(256) void k_fq_positions(const uint8_t* __restrict__ raw, uint64_t nbytes, const uint32_t* __restrict__ chunk_base,
                                                      uint32_t* __restrict__ nl, uint32_t cap) {
    __shared__ uint32_t wsum[4];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cr = 0, m = 0;
    if (i * 16 < nbytes) m = nl_mask16(load16(raw, nbytes, i), cr);
    const uint32_t n = (uint32_t)__popc(m);
    uint32_t incl = n; // inclusive scan over the wave
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = chunk_base[blockIdx.x] + incl - n;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
    const uint32_t b0 = (uint32_t)(i * 16);
    while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        if (before < cap) nl[before] = b0 + (uint32_t)bit;
        ++before;
    }
}

__device__ __forceinline__ bool is_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }

// info: [0] status bits, [1] records, [2] longest sequence, [3] (unused).  nlines_p = &chunk_base[nchunks] (the scan's total).
__global__ __launch_bounds__(256) void k_fq_records(const uint8_t* __restrict__ raw, uint64_t nbytes, const uint32_t* __restrict__ nl,
                                                    const uint32_t* __restrict__ nlines_p, uint32_t line_cap, uint32_t rec_cap,
                                                    uint32_t* __restrict__ seq_off, uint32_t* __restrict__ seq_len, uint32_t* __restrict__ qual_off,
                                                    uint32_t* __restrict__ name_off, uint32_t* __restrict__ name_len,
                                                    uint32_t* __restrict__ info) {
    const uint32_t nlines = *nlines_p;
    uint32_t bad = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (nlines > line_cap) bad |= FQ_BAD_CAP;
        if (nlines & 3u) bad |= FQ_BAD_LINES;                                   // lines come in fours
        if (nbytes && raw[nbytes - 1] != '\n') bad |= FQ_BAD_LINES;              // the block ends with a record's last newline
        if ((nlines >> 2) > rec_cap) bad |= FQ_BAD_CAP;
        info[1] = nlines >> 2;
    }
    const uint32_t nrec = nlines >> 2;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    uint32_t len = 0;
    if (r < nrec && nlines <= line_cap && nrec <= rec_cap) {
        const uint32_t s0 = r ? nl[4 * r - 1] + 1u : 0u;
        const uint32_t e0 = nl[4 * r], e1 = nl[4 * r + 1], e2 = nl[4 * r + 2], e3 = nl[4 * r + 3];
        const uint32_t s1 = e0 + 1u, s2 = e1 + 1u, s3 = e2 + 1u;
        len = e1 - s1;
        if (raw[s0] != '@' || raw[s2] != '+' || e3 - s3 != len) bad |= FQ_BAD_RECORD; // (an empty header line has raw[s0] == '\n')
        if (len == 0) bad |= FQ_BAD_RECORD; // kseq returns such a record after reading on for the next header: never guessed at
        uint32_t q = s0 + 1u;
        while (q < e0 && !is_space(raw[q])) ++q; // name = up to the first whitespace (kseq.hpp:181)
        seq_off[r] = s1; seq_len[r] = len; qual_off[r] = s3; name_off[r] = s0 + 1u; name_len[r] = q - (s0 + 1u);
    }
    if (bad) atomicOr(&info[0], bad);
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_down(len, o); len = v > len ? v : len; }
    if ((threadIdx.x & 63) == 0 && len) atomicMax(&info[2], len);
}

// one wave per record (grid-stride): the sequence bytes move to bases[out_off[r] ..), sequence and quality bytes are range-checked
__global__ __launch_bounds__(256) void k_fq_gather(const uint8_t* __restrict__ raw, const uint32_t* __restrict__ seq_off,
                                                   const uint32_t* __restrict__ seq_len, const uint32_t* __restrict__ qual_off,
                                                   const uint32_t* __restrict__ out_off, uint32_t* __restrict__ info,
                                                   uint8_t* __restrict__ bases) {
    if (info[0] != 0) return; // irregular block: nothing is trusted
    const uint32_t nrec = info[1];
    const uint32_t lane = threadIdx.x & 63, wpb = 4;
    uint32_t bad = 0;
    for (uint32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < nrec; r += gridDim.x * wpb) {
        const uint32_t s = seq_off[r], n = seq_len[r], o = out_off[r], p = qual_off[r];
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t c = raw[s + i], qc = raw[p + i];
            bases[o + i] = (uint8_t)c;
            if ((c - 33u) > 93u || c == '>' || c == '+' || c == '@') bad |= FQ_BAD_CHAR;
            if ((qc - 33u) > 94u) bad |= FQ_BAD_CHAR;
        }
    }
    if (bad) atomicOr(&info[0], bad);
}

// ---- what the host still needs of the text (launch_fastq_pack) ----
// pack_len[r] = bytes record r contributes; entries from the block's last record on are zeroed (the scan runs over the capacity)
__global__ __launch_bounds__(256) void k_fq_pack_sizes(const uint32_t* __restrict__ info, uint32_t rec_cap, const uint32_t* __restrict__ name_len,
                                                       const uint32_t* __restrict__ seq_len, const int32_t* __restrict__ out4, int names_only,
                                                       int min_matches, int min_diff, uint32_t* __restrict__ pack_len) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r > rec_cap) return;
    uint32_t n = 0;
    if (info[0] == 0 && r < info[1]) {
        if (names_only) n = name_len[r];
        else if (rk_filter_keeps(out4 + (size_t)r * 4, min_matches, min_diff)) n = name_len[r] + 2u * seq_len[r];
    }
    pack_len[r] = n;
}
// 16 lanes per record: name (and sequence, quality) -> pack[pack_off[r] ..)
__global__ __launch_bounds__(256) void k_fq_pack_copy(const uint8_t* __restrict__ raw, uint32_t* __restrict__ info, const uint32_t* __restrict__ name_off,
                                                      const uint32_t* __restrict__ name_len, const uint32_t* __restrict__ seq_off,
                                                      const uint32_t* __restrict__ seq_len, const uint32_t* __restrict__ qual_off,
                                                      const uint32_t* __restrict__ pack_len, const uint32_t* __restrict__ pack_off, int names_only,
                                                      uint8_t* __restrict__ pack, uint64_t pack_cap) {
    if (info[0] != 0) return;
    const uint32_t nrec = info[1];
    const uint32_t total = pack_off[nrec];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        info[3] = total;
        if ((uint64_t)total > pack_cap) atomicOr(&info[0], FQ_BAD_CAP);
    }
    if ((uint64_t)total > pack_cap) return;
    const uint32_t sub = threadIdx.x & 15u, per = 16;
    for (uint32_t r = blockIdx.x * per + (threadIdx.x >> 4); r < nrec; r += gridDim.x * per) {
        const uint32_t n = pack_len[r];
        if (n == 0) continue;
        uint8_t* const dst = pack + pack_off[r];
        const uint32_t nn = name_len[r];
        const uint8_t* src = raw + name_off[r];
        for (uint32_t i = sub; i < nn; i += 16) dst[i] = src[i];
        if (!names_only) {
            const uint32_t ns = seq_len[r];
            src = raw + seq_off[r];
            for (uint32_t i = sub; i < ns; i += 16) dst[nn + i] = src[i];
            src = raw + qual_off[r];
            for (uint32_t i = sub; i < ns; i += 16) dst[nn + ns + i] = src[i];
        }
    }
}

} // namespace

hipError_t launch_fastq_pack(const FqDev& d, const uint8_t* raw, bool names_only, const int32_t* out4, int min_matches, int min_diff, hipStream_t st) {
    hipLaunchKernelGGL(k_fq_pack_sizes, dim3((d.rec_cap + 256) / 256), dim3(256), 0, st, d.info, d.rec_cap, d.name_len, d.seq_len, out4, names_only ? 1 : 0,
                       min_matches, min_diff, d.pack_len);
    size_t tb = d.scan_tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.pack_len, d.pack_off, (int)(d.rec_cap + 1), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_pack_copy, dim3(4096), dim3(256), 0, st, raw, d.info, d.name_off, d.name_len, d.seq_off, d.seq_len, d.qual_off, d.pack_len,
                       d.pack_off, names_only ? 1 : 0, d.pack, d.pack_cap);
    return hipGetLastError();
}

size_t fq_scan_temp_bytes(uint32_t n) {
    size_t tb = 0;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, nullptr);
    (void)e;
    return tb + 256;
}

// raw: nbytes of FASTQ text on the device (16-byte aligned, readable up to the next multiple of 16).  Fills d.nl / spans /
// out_off ([rec_cap + 1], exclusive scan of the lengths: out_off[nrec] = total bases) / bases and d.info (status, records, longest).
hipError_t launch_fastq_index(const FqDev& d, const uint8_t* raw, uint64_t nbytes, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d.info, 0, 16, st);
    if (e != hipSuccess) return e;
    if (nbytes == 0) return hipSuccess;
    const uint32_t nchunks = (uint32_t)((nbytes + FQ_CHUNK - 1) / FQ_CHUNK);
    hipLaunchKernelGGL(k_fq_count, dim3(nchunks), dim3(256), 0, st, raw, nbytes, d.chunk_cnt, d.info);
    // chunk_cnt[nchunks] = 0 is part of the input so that the exclusive scan leaves the total there
    e = hipMemsetAsync(d.chunk_cnt + nchunks, 0, 4, st);
    if (e != hipSuccess) return e;
    size_t tb = d.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.chunk_cnt, d.chunk_base, (int)(nchunks + 1), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_positions, dim3(nchunks), dim3(256), 0, st, raw, nbytes, d.chunk_base, d.nl, d.line_cap);
    hipLaunchKernelGGL(k_fq_records, dim3((d.rec_cap + 255) / 256), dim3(256), 0, st, raw, nbytes, d.nl, d.chunk_base + nchunks, d.line_cap,
                       d.rec_cap, d.seq_off, d.seq_len, d.qual_off, d.name_off, d.name_len, d.info);
    // (the scan runs over the whole capacity: entries past the block's last record hold stale lengths, which an exclusive scan never
    // lets reach out_off[0 .. records])
    tb = d.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(d.scan_tmp, tb, d.seq_len, d.out_off, (int)(d.rec_cap + 1), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fq_gather, dim3(2048), dim3(256), 0, st, raw, d.seq_off, d.seq_len, d.qual_off, d.out_off, d.info, d.bases);
    return hipGetLastError();
}

hipError_t warm_inflate(); // rk_inflate.hip
static const void* k_fq_count_for_warm_up() { return reinterpret_cast<const void*>(k_fq_count); }

} // namespace rk

// Loads the code objects of the device front end's kernels (and, with_inflate != 0, of the device inflater) for the CURRENT device
// ahead of their first launch: ~50 ms (150 ms with the inflater) that a caller can spend on a second thread while its
// references are sketched, instead of in front of its first block.  Thread-safe; harmless to repeat.
extern "C" int rk_warm_up(int device, int with_inflate) {
    if (hipSetDevice(device) != hipSuccess) return -3;
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, rk::k_fq_count_for_warm_up());
    if (e == hipSuccess && with_inflate) e = rk::warm_inflate();
    return e == hipSuccess ? 0 : -3;
}
