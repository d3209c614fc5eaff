// rk_inflate.hip -- DEFLATE (RFC 1951) on the device, for BGZF members (gfx950, wave64).
//
// The reference opens every input with gzopen (/root/reference/src/rkmh.cpp:238-263): one sequential inflater.  A BGZF file is a chain
// of INDEPENDENT gzip members of at most 64 KB of text each (SAM specification 4.1), thousands per block of the device FASTQ front
// end -- so the block's compressed bytes cross the link (0.58 x the text for level-1 FASTQ) and ONE WAVE PER MEMBER inflates them:
//   * TWO PASSES.  Decoding the Huffman symbols needs the bit stream and the code tables, not the text: pass 1 (k_inflate_members) decodes
//     with 8 KB of LDS per wave -- five waves per SIMD hide each other's table and refill latencies --, stores every literal at its final
//     place in HBM and appends every match (position, length, distance) to the member's list; pass 2 (k_inflate_resolve) loads the
//     member's text into a 64 KB LDS window, applies the matches in order (64 lanes per copy) and writes the member back.  A first
//     version did both in one pass with the window in LDS: two waves per CU, 2.6 GB/s of text (profiles/r05_gz.txt);
//   * the Huffman decode is the wave's serial part, executed uniformly by all lanes (no divergence, LDS table reads are
//     broadcasts): one 10-bit lookup per symbol for codes of up to 10 bits, the canonical count / offset walk for the rare longer ones;
//   * the compressed stream lives in registers, 8 bytes per lane (512 bytes per wave, the next 512 already requested): the bit
//     reader refills with two v_readlane, never from memory.
// CRC-32 is NOT checked here (ISIZE and the stream's own end-of-block structure are); a damaged member almost surely breaks the
// four-line grammar that the front end verifies next, and RKMH_BGZF_DEVICE=0 keeps the host inflater with its CRC check.
#include "rk_kernels.hpp"

namespace rk {

namespace {

constexpr int IW = 64;
constexpr int LIT_BITS = 10, DIST_BITS = 9;
constexpr uint32_t WIN_BYTES = 65536;

// table entry: bits 0..3 code length (0: longer than the table's bits -> canonical walk), 4..5 kind, 8.. payload
//   lit/len kind 0 literal (payload = byte), 1 length (payload = base | extra << 9), 2 end of block, 3 invalid
//   dist    payload = base | extra << 16
struct InfLds {
    uint32_t lit[1 << LIT_BITS];
    uint32_t dist[1 << DIST_BITS];
    uint16_t lsym[288], dsym[32];   // symbols sorted by (code length, symbol): the canonical walk for long codes
    uint16_t lcount[16], dcount[16];
    uint8_t lens[320];
    uint32_t clt[128];              // code-length code table (<= 7 bits)
};

__device__ const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ void isync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t bitrev(uint32_t v, int n) { return __builtin_bitreverse32(v) >> (32 - n); }

// the compressed stream of one member in registers: A = bytes [base, base + 512), B = the next 512, 8 bytes per lane
struct InStream {
    const uint8_t* src;   // 8-byte aligned start (at or below the member's first payload byte)
    uint32_t limit;       // bytes readable from src (whole 8-byte words inside the block's compressed buffer)
    uint32_t base;        // offset of A in src
    uint2 a, b;
    uint32_t pos;         // next unread byte, relative to src
    uint64_t bb;          // bit buffer
    uint32_t nb;          // valid bits in bb
    int lane;
    // (the load is waited for HERE: with a load in flight the compiler's waits inside the symbol loop would also wait for every
    // literal store before them -- a microsecond per symbol; a chunk is 512 bytes of input, about a kilobyte of text)
    __device__ __forceinline__ uint2 load_chunk(uint32_t off) const {
        const uint32_t o = off + 8u * (uint32_t)lane;
        uint2 v = make_uint2(0u, 0u);
        if (o + 8u <= limit) v = *reinterpret_cast<const uint2*>(src + o);
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        return v;
    }
    __device__ __forceinline__ void start(const uint8_t* s, uint32_t lim, uint32_t first, int ln) {
        src = s; limit = lim; lane = ln; base = first & ~511u;
        a = load_chunk(base); b = load_chunk(base + 512u);
        pos = first; bb = 0; nb = 0;
    }
    __device__ __forceinline__ uint32_t dword(uint32_t w) const { // dword w of A|B (0 .. 255), w wave-uniform
        const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)((w >> 1) & 63u));
        const bool hi = (w & 1u) != 0u, second = w >= 128u;
        const uint32_t va = hi ? a.y : a.x, vb = hi ? b.y : b.x;
        return (uint32_t)__builtin_amdgcn_readlane((int)(second ? vb : va), (int)l);
    }
    __device__ __forceinline__ void refill() { // afterwards nb >= 33 (while input lasts)
        if (nb <= 32u) {
            if (pos - base >= 512u) { a = b; base += 512u; b = load_chunk(base + 512u); }
            const uint32_t r = pos - base, w = r >> 2, sh = (r & 3u) * 8u;
            const uint32_t d0 = dword(w), d1 = dword(w + 1u);
            const uint32_t v = sh ? (d0 >> sh) | (d1 << (32u - sh)) : d0;
            bb |= (uint64_t)v << nb;
            pos += 4u; nb += 32u;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; nb -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    // byte position of the next unread BIT's byte after aligning to a byte boundary
    __device__ __forceinline__ uint32_t align_to_byte() { drop(nb & 7u); const uint32_t p = pos - (nb >> 3); bb = 0; nb = 0; return p; }
    __device__ __forceinline__ void seek(uint32_t p) { pos = p; bb = 0; nb = 0; if (p - base >= 1024u || p < base) { base = p & ~511u; a = load_chunk(base); b = load_chunk(base + 512u); } }
};

// canonical Huffman tables from code lengths lens[0 .. n): table of 2^TB entries for codes of <= TB bits, count / sorted symbols for longer ones
template <int TB, bool DIST>
__device__ bool build_table(const uint8_t* lens, int n, uint32_t* tab, uint16_t* count, uint16_t* sorted, int lane) {
    if (lane < 16) count[lane] = 0;
    isync();
    if (lane == 0) for (int s = 0; s < n; ++s) count[lens[s]]++;
    isync();
    // over-subscribed code -> invalid; (incomplete codes are legal for a single distance code)
    int left = 1;
    uint32_t first[16], offs[16];
    uint32_t code = 0, o = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - (int)count[l];
        if (left < 0) return false;
        first[l] = code; offs[l] = o;
        code = (code + count[l]) << 1; o += count[l];
    }
    for (int i = lane; i < (1 << TB); i += IW) tab[i] = 0x3u << 4; // invalid
    isync();
    // symbols in order: code(sym) = first[len]++ ; entry replicated over the unused high index bits
    uint32_t next[16], noff[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) { next[l] = l ? first[l] : 0u; noff[l] = l ? offs[l] : 0u; }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l == 0) continue;
        uint32_t c = 0, so = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) if (q == l) { c = next[q]++; so = noff[q]++; }
        if (lane == 0) sorted[so] = (uint16_t)s;
        if (l > TB) {
            if (lane == 0) tab[bitrev(c >> (l - TB), TB)] = 0u; // (prefix of a long code: length 0 = walk)
            continue;
        }
        uint32_t e;
        if (DIST) e = (uint32_t)l | ((uint32_t)DIST_BASE[s < 30 ? s : 0] << 8) | ((uint32_t)DIST_EXTRA[s < 30 ? s : 0] << 24) | (s >= 30 ? 0x30u : 0u);
        else if (s < 256) e = (uint32_t)l | ((uint32_t)s << 8);
        else if (s == 256) e = (uint32_t)l | (2u << 4);
        else if (s < 286) e = (uint32_t)l | (1u << 4) | ((uint32_t)LEN_BASE[s - 257] << 8) | ((uint32_t)LEN_EXTRA[s - 257] << 17);
        else e = (uint32_t)l | (3u << 4);
        const uint32_t r = bitrev(c, l);
        for (uint32_t j = (uint32_t)lane; j < (1u << (TB - l)); j += IW) tab[r | (j << l)] = e;
    }
    isync();
    return true;
}

// a symbol whose code is longer than the table's bits: the canonical walk (puff.c), bit by bit from the stream
__device__ __forceinline__ int walk_long(InStream& in, const uint16_t* count, const uint16_t* sorted) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)in.take(1);
        const int c = count[l];
        if (code - c < first) return sorted[index + (code - first)];
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

} // namespace

// status[m]: 0 ok, else the member could not be inflated (nothing of the job is used)
__global__ __launch_bounds__(IW) void k_inflate_members(const uint8_t* __restrict__ comp, uint32_t comp_bytes, const InflateMember* __restrict__ mem, uint32_t nmem,
                                                        uint8_t* __restrict__ text, uint2* __restrict__ matches, uint32_t* __restrict__ status) {
    __shared__ InfLds L;
    const int lane = threadIdx.x;
    const uint32_t m = blockIdx.x;
    if (m >= nmem) return;
    const InflateMember me = mem[m];
    uint32_t bad = 0;
    uint32_t op = 0, nmatch = 0;
    uint8_t* const dst = text + me.out_off;
    uint2* const ml = matches + me.match_off;
    const uint32_t match_cap = me.out_len / 3u + 1u; // (a match is at least three bytes)
    if (me.out_len > WIN_BYTES) bad = 1;
    InStream in;
    in.start(comp, comp_bytes & ~7u, me.in_off, lane);
    const uint32_t in_end = me.in_off + me.in_len;
    bool final_block = false;
    while (!bad && !final_block) {
        in.refill();
        final_block = in.take(1) != 0u;
        const uint32_t type = in.take(2);
        if (type == 0u) { // stored
            uint32_t p = in.align_to_byte();
            if (p + 4u > in_end) { bad = 2; break; }
            const uint32_t len = (uint32_t)comp[p] | ((uint32_t)comp[p + 1] << 8), nlen = (uint32_t)comp[p + 2] | ((uint32_t)comp[p + 3] << 8);
            p += 4u;
            if ((len ^ 0xFFFFu) != nlen || p + len > in_end || op + len > me.out_len) { bad = 3; break; }
            for (uint32_t i = (uint32_t)lane; i < len; i += IW) dst[op + i] = comp[p + i];
            op += len;
            in.seek(p + len);
            continue;
        }
        if (type == 3u) { bad = 4; break; }
        int nlit = 288, ndist = 30;
        if (type == 1u) { // fixed codes
            for (int i = lane; i < 288; i += IW) L.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
            if (lane < 30) L.lens[288 + lane] = 5;
            isync();
        } else { // dynamic: code length code, then the literal/length and distance code lengths
            in.refill();
            nlit = (int)in.take(5) + 257; ndist = (int)in.take(5) + 1;
            const int ncl = (int)in.take(4) + 4;
            if (nlit > 286 || ndist > 30) { bad = 5; break; }
            if (lane < 19) L.lens[lane] = 0;
            isync();
            for (int i = 0; i < ncl; ++i) { in.refill(); const uint32_t v = in.take(3); if (lane == 0) L.lens[CL_ORDER[i]] = (uint8_t)v; }
            isync();
            // (the code-length table reuses the literal builder: symbols 0..18 as "literals")
            if (!build_table<7, false>(L.lens, 19, L.clt, L.lcount, L.lsym, lane)) { bad = 6; break; }
            int i = 0;
            uint32_t prev = 0;
            // the lengths go to lens[32 ..) while lens[0..19) still holds the code-length code
            uint8_t* out = L.lens;
            uint8_t tmp_prev = 0;
            (void)tmp_prev;
            // decode into registers-free LDS area: first pass writes to a second array -- the builder's input is read before it is
            // overwritten because the table above is complete
            while (i < nlit + ndist && !bad) {
                in.refill();
                const uint32_t e = L.clt[in.peek(7)];
                const uint32_t l = e & 15u;
                if (l == 0u || ((e >> 4) & 3u) == 3u) { bad = 7; break; }
                in.drop(l);
                const uint32_t sym = e >> 8;
                if (sym < 16u) { if (lane == 0) out[i] = (uint8_t)sym; prev = sym; ++i; }
                else {
                    uint32_t rep, val = 0;
                    if (sym == 16u) { if (i == 0) { bad = 8; break; } rep = 3u + in.take(2); val = prev; }
                    else if (sym == 17u) { rep = 3u + in.take(3); prev = 0; }
                    else { rep = 11u + in.take(7); prev = 0; }
                    if (i + (int)rep > nlit + ndist) { bad = 9; break; }
                    for (uint32_t j = (uint32_t)lane; j < rep; j += IW) out[i + (int)j] = (uint8_t)val;
                    i += (int)rep;
                }
            }
            if (bad) break;
            isync();
            if (L.lens[256] == 0) { bad = 10; break; }
        }
        if (!build_table<LIT_BITS, false>(L.lens, nlit, L.lit, L.lcount, L.lsym, lane)) { bad = 11; break; }
        if (!build_table<DIST_BITS, true>(L.lens + nlit, ndist, L.dist, L.dcount, L.dsym, lane)) { bad = 12; break; }
        // ---- symbols ----
        for (;;) {
            in.refill();
            uint32_t e = L.lit[in.peek(LIT_BITS)];
            uint32_t l = e & 15u, kind = (e >> 4) & 3u, payload = e >> 8;
            if (l == 0u) { // a code longer than the table's bits (or an unused prefix)
                if (kind == 3u) { bad = 13; break; }
                const int s = walk_long(in, L.lcount, L.lsym);
                if (s < 0 || s >= 286) { bad = 14; break; }
                if (s < 256) { kind = 0; payload = (uint32_t)s; }
                else if (s == 256) kind = 2;
                else { kind = 1; payload = (uint32_t)LEN_BASE[s - 257] | ((uint32_t)LEN_EXTRA[s - 257] << 9); }
            } else {
                if (kind == 3u) { bad = 15; break; }
                in.drop(l);
            }
            if (kind == 0u) {
                if (op >= me.out_len) { bad = 16; break; }
                if (lane == 0) dst[op] = (uint8_t)payload;
                ++op;
                continue;
            }
            if (kind == 2u) break;
            const uint32_t eb = payload >> 9;
            in.refill();
            const uint32_t len = (payload & 511u) + in.take(eb);
            uint32_t de = L.dist[in.peek(DIST_BITS)];
            uint32_t dl = de & 15u, dbase, dext;
            if (dl == 0u) {
                if (((de >> 4) & 3u) == 3u) { bad = 17; break; }
                const int s = walk_long(in, L.dcount, L.dsym);
                if (s < 0 || s >= 30) { bad = 18; break; }
                dbase = DIST_BASE[s]; dext = DIST_EXTRA[s];
            } else {
                if (((de >> 4) & 3u) == 3u) { bad = 19; break; }
                in.drop(dl);
                dbase = (de >> 8) & 0xFFFFu; dext = de >> 24;
            }
            in.refill();
            const uint32_t dist = dbase + in.take(dext);
            if (dist > op || op + len > me.out_len || nmatch >= match_cap) { bad = 20; break; }
            if (lane == 0) ml[nmatch] = make_uint2(op, len | (dist << 16)); // resolved by pass 2, in order
            ++nmatch;
            op += len;
        }
    }
    if (!bad && op != me.out_len) bad = 21;
    if (lane == 0) { status[m] = bad; status[nmem + m] = bad ? 0u : nmatch; }
}

// pass 2: the member's matches applied in order inside a 64 KB LDS window (a match may copy what an earlier match wrote)
__global__ __launch_bounds__(IW) void k_inflate_resolve(const InflateMember* __restrict__ mem, uint32_t nmem, uint8_t* __restrict__ text, const uint2* __restrict__ matches,
                                                        const uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t win[WIN_BYTES];
    const int lane = threadIdx.x;
    const uint32_t m = blockIdx.x;
    if (m >= nmem) return;
    const uint32_t nmatch = status[nmem + m];
    if (status[m] != 0u || nmatch == 0u) return; // (no matches: the literals are the text)
    const InflateMember me = mem[m];
    uint8_t* const dst = text + me.out_off;
    const uint32_t n = me.out_len;
    // the text so far (literals in place, holes where matches go): 4 bytes per lane where the address allows
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) < n ? (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) : n;
    const uint32_t nd = (n - head) >> 2, done = head + 4u * nd;
    if ((uint32_t)lane < head) win[lane] = dst[lane];
    for (uint32_t i = (uint32_t)lane; i < nd; i += IW) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(dst + head + 4u * i);
        const uint32_t o = head + 4u * i;
        win[o] = (uint8_t)v; win[o + 1] = (uint8_t)(v >> 8); win[o + 2] = (uint8_t)(v >> 16); win[o + 3] = (uint8_t)(v >> 24);
    }
    if ((uint32_t)lane < n - done) win[done + lane] = dst[done + lane];
    isync();
    const uint2* ml = matches + me.match_off;
    for (uint32_t q0 = 0; q0 < nmatch; q0 += IW) { // 64 entries per load (one per lane), handed round by v_readlane
        uint2 mine = make_uint2(0u, 0u);
        if (q0 + (uint32_t)lane < nmatch) mine = ml[q0 + (uint32_t)lane];
        const uint32_t cnt = nmatch - q0 < (uint32_t)IW ? nmatch - q0 : (uint32_t)IW;
        for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)__builtin_amdgcn_readfirstlane((int)j));
            const uint32_t ld = (uint32_t)__builtin_amdgcn_readlane((int)mine.y, (int)__builtin_amdgcn_readfirstlane((int)j));
            const uint32_t len = ld & 0xFFFFu, dist = ld >> 16;
            if (dist >= len || dist >= (uint32_t)IW) {
                for (uint32_t i0 = 0; i0 < len; i0 += IW) { // (dist >= 64: a step's 64 source bytes were all written before it)
                    const uint32_t i = i0 + (uint32_t)lane;
                    uint8_t v = 0;
                    if (i < len) v = win[op - dist + i];
                    isync();
                    if (i < len) win[op + i] = v;
                    isync();
                }
            } else { // the source overlaps the destination: it repeats with period dist
                for (uint32_t i = (uint32_t)lane; i < len; i += IW) win[op + i] = win[op - dist + i % dist];
                isync();
            }
        }
    }
    if ((uint32_t)lane < head) dst[lane] = win[lane];
    for (uint32_t i = (uint32_t)lane; i < nd; i += IW) {
        const uint32_t o = head + 4u * i;
        *reinterpret_cast<uint32_t*>(dst + o) = (uint32_t)win[o] | ((uint32_t)win[o + 1] << 8) | ((uint32_t)win[o + 2] << 16) | ((uint32_t)win[o + 3] << 24);
    }
    if ((uint32_t)lane < n - done) dst[done + lane] = win[done + lane];
}

// first record start (four-line rule, as find_record_start in rk_parse.cpp) at or after `from` in text[0 .. n): a line start p with
// text[p] == '@' whose line two below begins with '+'.  One workgroup; cuts[which] = the position, n if there is none (the text
// ends inside the last record's lines), 0xFFFFFFFF if the lookahead ran out of text (the caller inflates more members).
__global__ __launch_bounds__(256) void k_fastq_first_start(const uint8_t* __restrict__ text, uint32_t n, uint32_t from, uint32_t window, uint32_t at_eof,
                                                           uint32_t* __restrict__ cuts, int which) {
    __shared__ uint32_t best, starved;
    if (threadIdx.x == 0) { best = 0xFFFFFFFFu; starved = 0; }
    __syncthreads();
    const uint32_t hi = from + window < n ? from + window : n;
    for (uint32_t p = from + threadIdx.x; p < hi; p += blockDim.x) {
        if (text[p] != '@' || (p > 0 && text[p - 1] != '\n')) continue;
        if (p >= best) break;
        uint32_t q = p, nl = 0;
        while (q < n && nl < 2) { if (text[q] == '\n') ++nl; ++q; }
        if (nl < 2 || q >= n) { atomicMax(&starved, 1u); continue; }
        if (text[q] == '+') atomicMin(&best, p);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = best;
        if (r == 0xFFFFFFFFu) r = (hi == n && (at_eof || !starved)) ? (at_eof ? n : 0xFFFFFFFFu) : 0xFFFFFFFFu;
        cuts[which] = r;
    }
}

hipError_t launch_inflate_members(const uint8_t* comp, uint32_t comp_bytes, const InflateMember* mem, uint32_t nmem, uint8_t* text, uint2* matches, uint32_t* status,
                                  hipStream_t st) {
    if (!nmem) return hipSuccess;
    hipLaunchKernelGGL(k_inflate_members, dim3(nmem), dim3(IW), 0, st, comp, comp_bytes, mem, nmem, text, matches, status);
    hipLaunchKernelGGL(k_inflate_resolve, dim3(nmem), dim3(IW), 0, st, mem, nmem, text, matches, status);
    return hipGetLastError();
}
hipError_t launch_fastq_first_start(const uint8_t* text, uint32_t n, uint32_t from, uint32_t window, bool at_eof, uint32_t* cuts, int which, hipStream_t st) {
    hipLaunchKernelGGL(k_fastq_first_start, dim3(1), dim3(256), 0, st, text, n, from, window, at_eof ? 1u : 0u, cuts, which);
    return hipGetLastError();
}

} // namespace rk
