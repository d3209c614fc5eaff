// rk_inflate.hip -- DEFLATE (RFC 1951) on the device: BGZF members, and the chunks of ordinary gzip streams (gfx950, wave64).
//
// The reference opens every input with gzopen (/root/reference/src/rkmh.cpp:238-263): one sequential inflater.  A BGZF file is a chain
// of INDEPENDENT gzip members of at most 64 KB of text each (SAM specification 4.1), about a thousand per block of the device FASTQ
// front end -- so the block's compressed bytes cross the link (0.58 x the text for level-1 FASTQ) and the device inflates them:
//   * pass 1, k_inflate_lanes: ONE LANE PER MEMBER.  Huffman decoding is a serial chain per stream; a wave that decodes one stream
//     uniformly spends 64 lanes on it (the first version here: ~800 cycles per symbol, the chip's issue slots full at 3 GB/s of
//     text, profiles/r05_gz.txt).  Here every lane runs its own stream: bit buffer, position and block state in registers, its code
//     tables (8-bit literal/length root, 6-bit distance root, 16-bit entries), its input ring and its output rings in LDS, all
//     arrays interleaved by lane.  What a lane produces is NOT text but (a) the member's literals as one packed byte stream and (b) a
//     32-bit entry per match: run of literals before it (8 bits) | length (9) | distance - 1 (15); an entry of length 0 carries 255
//     literals of a longer run, or the tail.
//   * the wave's memory traffic is wave-uniform: a wait for a load is a wait for the whole wave, so no lane ever loads or stores
//     inside the symbol loop.  Once per PERIOD (K_SYM symbol steps) every lane takes the 16 input bytes it requested one period
//     earlier into its ring, flushes whole literal dwords and entries, and requests its next 16 bytes; a lane whose ring runs low or
//     whose output rings fill idles until the next period.  Block headers are decoded a few steps per period by the lanes that are
//     at one; a lane that needs code tables asks for them and the WAVE builds them (64 symbols at a time: codes by ballot ranks);
//   * pass 2, k_inflate_place: one wave per member and a 35 KB wrapping LDS window (see the kernel);
//   * k_crc32_members: every member's CRC-32 against its footer (what gzread checks); a mismatch hands the job to the host inflater.
// The STREAM forms of the two passes, k_gz_find_starts, k_gz_windows and k_gz_resolve inflate ORDINARY gzip files -- one deflate
// stream -- from block headers found in the stream (rk_gunzip.hip drives them; the section further down explains how).
#include "rk_kernels.hpp"
#include "rk_crc32.hpp"
#include <atomic>
#include <cstdlib>

namespace rk {

namespace {

constexpr int IW = 64;
constexpr int CT = 7;                         // root table bits of the code-length code
// Root table bits of the literal/length and the distance code are template parameters of the kernel (LT, DT).  The shipped form is
// <8, 6>: 73.5 KB of LDS per wave, so TWO waves decode on every CU (160 KB) -- a wave is issue-latency bound on its own SIMD and the
// other three SIMDs of the CU idle, so the second wave is almost free (round 5 ran <9, 7>: 124.5 KB, one wave per CU, 26.6 ms per
// launch whatever its size).  The price: 83 % instead of 62 % of the symbol windows meet a code longer than the root: 30.3 ms per wave.
constexpr int LONG_CAP = 128, DLONG_CAP = 32;  // symbols with codes longer than the root (more: the member is the host's)
constexpr int WALK_N = 10;                    // code lengths above the root's bits (15 - 6 = 9 at most)
constexpr int CL_AT = 320, LENS_N = 352;      // lens[0 .. 316): literal/length + distance code lengths; lens[320 .. 339): code-length code
constexpr int IN_RING = 16, ENT_RING = 16, OUT_RING = 4; // dwords per lane
constexpr int K_SYM = 8, K_HDR = 8;           // symbol steps / header steps (inside a header window) per period
constexpr uint32_t HDR_EVERY = 64;            // periods between header windows (a power of two)
constexpr int HDR_LANES = 12;                 // ... unless this many lanes wait
constexpr uint32_t E_INVALID = 0x0FFFu;       // table entry: code length << 12 | symbol; length 0: E_INVALID, or a long code's prefix
constexpr uint32_t WIN_BYTES = 65536;       // a member's text at most (pass 1 hands larger ones to the host)

__device__ const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
constexpr uint8_t CL_ORDER_C[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; // (the same, for unrolled loops)

// everything a wave of 64 streams keeps in LDS; arrays are [index][lane]
template <int LT, int DT>
struct LaneLds {
    uint16_t lit[(1 << LT) * IW];
    uint16_t dist[(1 << DT) * IW];
    // lsort / dsort: long-coded symbols in (length, symbol) order: the canonical walk.  A stream's code lengths (lens: 0 .. 15,
    // four per 16-bit word) are only alive between its block header and the table build, its sorted long symbols from the build to
    // the block's end -- the same column of the same lane serves both (the build reads every length before it writes a symbol)
    union {
        uint16_t lsort[LONG_CAP * IW];
        uint16_t lens[LENS_N / 4 * IW];
    };
    uint16_t dsort[DLONG_CAP * IW];
    // per code length above the root's bits: left-justified 15-bit upper limit of its codes | (index in the sorted list - first code) << 16
    uint32_t lwalk[WALK_N * IW], dwalk[WALK_N * IW];
    uint32_t inr[IN_RING * IW];
    uint32_t entr[(ENT_RING + 1) * IW]; // (one more row each: where a step's store goes that is not due)
    uint32_t outr[(OUT_RING + 1) * IW];
    uint8_t clorder[32];
};
static_assert(LENS_N / 4 <= LONG_CAP, "the code lengths live in the sorted list's rows");

template <class LDS> __device__ __forceinline__ uint32_t len_at(const LDS& L, int i, int t) { return ((uint32_t)L.lens[(i >> 2) * IW + t] >> (4 * (i & 3))) & 15u; }
// (a stream's column is written by its own lane only -- or by the whole wave, a word per lane, when the wave fills it)
template <class LDS> __device__ __forceinline__ void set_len(LDS& L, int i, int t, uint32_t v) {
    uint16_t& w = L.lens[(i >> 2) * IW + t];
    const uint32_t sh = 4u * ((uint32_t)i & 3u);
    w = (uint16_t)(((uint32_t)w & ~(15u << sh)) | (v << sh));
}

enum : uint32_t { ST_BLOCK = 0, ST_STORED_HDR, ST_DYN_HDR, ST_CL_READ, ST_LENS, ST_WAIT, ST_SYM, ST_STORED, ST_DONE, ST_FIN };

__device__ __forceinline__ void isync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t n) { return __builtin_bitreverse32(v) >> (32u - n); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t m, int lane) { return (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); }
// A member's part of the scratch buffer, in dwords.  Entries (one dword each) grow UP from its first dword, the literal stream (four
// literals per dword) grows DOWN from its last: a match entry stands for >= 3 bytes of text, a literal for one, so the two together
// never need more than out_len / 3 + 2 dwords (all matches of length 3: out_len / 3 entries, no literals; every literal more takes a
// byte of text from the matches) -- 1.34 bytes per byte of text instead of the 2.35 of two separate worst-case regions.
__device__ __host__ __forceinline__ uint32_t scratch_dwords(uint32_t out_len) { return out_len / 3u + 8u; }

// The wave builds stream t's canonical Huffman table from lens[(at + s) * 64 + t], s < n: root table of 2^TB entries, counts per
// length above the root (limit | offset, see LaneLds), the long-coded symbols in canonical order.  NCH = chunks of 64 symbols.  False:
// over-subscribed, more long codes than cap, or incomplete where zlib refuses that (inftrees.c: an incomplete set is accepted only
// for a literal/length or distance code whose longest code has one bit -- or none at all; never for the code-length code).
// l[c] = the code length of symbol lane + 64 c (the build's input, taken before anything of the build is written: the sorted list shares the lengths' rows)
template <int NCH, class LDS>
__device__ __forceinline__ void load_lens(const LDS& L, int t, int at, int n, int lane, uint32_t (&l)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int s = lane + IW * c;
        l[c] = s < n ? len_at(L, at + s, t) : 0u;
    }
}
template <int TB, int NCH>
__device__ bool coop_build(const uint32_t (&l)[NCH], int t, uint16_t* tab, uint16_t* sorted, int cap, uint32_t* walk, int lane, bool is_codes, bool may_be_incomplete = false) {
    uint32_t code_of[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) code_of[c] = 0;
    for (int i = lane; i < (1 << TB); i += IW) tab[i * IW + t] = (uint16_t)E_INVALID;
    if (walk && lane < WALK_N) walk[lane * IW + t] = 0; // (limit 0: no code of that length)
    uint32_t code = 0, longbase = 0, maxlen = 0;
    int left = 1;
    bool ok = true;
    for (uint32_t len = 1; len <= 15; ++len) {
        uint32_t n_len = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint64_t m = __ballot(l[c] == len);
            if (l[c] == len) code_of[c] = code + n_len + lanes_below(m, lane);
            n_len += (uint32_t)__popcll(m);
        }
        left = (left << 1) - (int)n_len;
        if (left < 0) ok = false;
        if (n_len) maxlen = len;
        if (len > (uint32_t)TB) {
            if (longbase + n_len > (uint32_t)cap) ok = false;
            else if (n_len) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (l[c] == len) sorted[(longbase + code_of[c] - code) * IW + t] = (uint16_t)(lane + IW * c);
            }
            if (lane == 0 && walk) walk[(len - TB - 1) * IW + t] = ((code + n_len) << (15u - len)) | (((longbase - code) & 0xFFFFu) << 16);
            longbase += n_len;
        }
        code = (code + n_len) << 1;
    }
    if (left > 0 && maxlen != 0u && (is_codes || maxlen != 1u) && !may_be_incomplete) ok = false;
    isync(); // (the invalid fill is in the table before the entries)
    if (ok) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t len = l[c];
            if (len == 0u) continue;
            if (len <= (uint32_t)TB) {
                const uint32_t e = (len << 12) | (uint32_t)(lane + IW * c);
                for (uint32_t j = bitrev(code_of[c], len); j < (1u << TB); j += 1u << len) tab[j * IW + t] = (uint16_t)e;
            } else {
                tab[bitrev(code_of[c] >> (len - TB), TB) * IW + t] = 0; // a long code's prefix: length 0, not E_INVALID
            }
        }
    }
    isync();
    return ok;
}

struct __attribute__((packed, aligned(4))) Quad { uint32_t v[4]; };

} // namespace

// Pass 1.  scratch: per member at dword me.match_off: scratch_dwords(out_len) dwords -- entries from the front, the literal stream from the back.
// status[m]: 0 ok, else why the member is the host's; status[nmem + m] = its entries | its literals << 15.
// STREAM: the lanes decode CHUNKS of one long deflate stream (an ordinary .gz file, rk_gunzip.hip) instead of members: a chunk begins at a
// block header anywhere in the stream (any bit), ends at the first block boundary at or after its stop position (or with the
// stream's last block), may copy from text in front of it (the 32 KB before its first byte: pass 2 knows what to do), and has no
// ISIZE -- its part of the scratch buffer bounds it instead.  mem is then a GzChunk array, which also takes the results.
template <int LT, int DT, bool STREAM>
__global__ __launch_bounds__(IW) void k_inflate_lanes(const uint8_t* __restrict__ comp, uint32_t comp_bytes, const void* __restrict__ mem_, uint32_t nmem,
                                                      uint32_t* __restrict__ scratch, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t inflate_lds[];
    typedef LaneLds<LT, DT> Lds;
    Lds& L = *reinterpret_cast<Lds*>(inflate_lds);
    const int lane = threadIdx.x;
    const uint32_t m = blockIdx.x * IW + (uint32_t)lane;
    const bool live = m < nmem;
    if (lane < 19) L.clorder[lane] = CL_ORDER[lane];
    InflateMember me = {0, 0, 0, 0, 0, 0};
    uint32_t stop_bit = 0, region_dw = 0, start_skip = 0, end_bit = 0;
    bool no_text_before = true;
    if constexpr (STREAM) {
        if (live) {
            const GzChunk g = reinterpret_cast<const GzChunk*>(mem_)[m];
            me.in_off = g.start_bit >> 3; start_skip = g.start_bit & 7u;
            me.out_len = 0xFFFFFFFFu; // (no ISIZE: the scratch region is the bound)
            me.match_off = g.scratch_off; region_dw = g.scratch_dw; stop_bit = g.stop_bit;
            no_text_before = (g.flags & 1u) != 0u;
        }
    } else {
        if (live) me = reinterpret_cast<const InflateMember*>(mem_)[m];
    }
    const uint32_t w0 = me.in_off >> 2;
    const uint32_t* const src = reinterpret_cast<const uint32_t*>(comp) + w0;
    uint32_t in_lim = live ? (STREAM ? 0xFFFFFFF0u : ((me.in_off & 3u) + me.in_len + 32u + 3u) >> 2) : 0u; // dwords of this stream worth loading (footer and a little more: the caller's buffer has 64 bytes of slack)
    { const uint32_t have = (comp_bytes >> 2) > w0 ? (comp_bytes >> 2) - w0 : 0u; if (in_lim > have) in_lim = have; }
    in_lim &= ~3u; // (whole 16-byte requests; the 32 bytes above leave the deflate data and two more dwords inside them)
    uint32_t* const ents = scratch + me.match_off;
    uint32_t* const lits_top = ents + ((STREAM ? region_dw : scratch_dwords(me.out_len)) - 1u); // literal dword d lives at lits_top[-d]
    uint64_t bb = 0;
    uint32_t nb = 0, in_r = 0, in_w = 0;
    uint32_t state = live ? ST_BLOCK : ST_FIN, bad = 0, need_build = 0;
    if (!STREAM && live && me.out_len > WIN_BYTES) { bad = 1; state = ST_FIN; }
    if (STREAM && live && region_dw < 64u) { bad = 22; state = ST_FIN; }
    uint32_t op = 0, run = 0, lit_n = 0, lit_acc = 0, out_f = 0, out_target = 0, ent_w = 0, ent_f = 0;
    uint32_t final_block = 0, nlit = 0, ndist = 0, ncl = 0, hi = 0, prev = 0, stored_left = 0;
    Quad pend = {{0, 0, 0, 0}};
    bool pending = live && in_lim >= 4u;
    if (pending) pend = *reinterpret_cast<const Quad*>(src);
    isync();

    // the bit buffer is refilled from two dwords held in registers (taken from the ring one step ahead: no LDS latency in the chain)
    uint32_t nx0 = 0, nx1 = 0, nav = 0;
    auto refill = [&]() {
        if (nb <= 32u && nav) { bb |= (uint64_t)nx0 << nb; nb += 32u; nx0 = nx1; --nav; }
    };
    auto topup = [&]() {
        if (nav < 2u && in_r < in_w) {
            const uint32_t at = (in_r & (IN_RING - 1)) * IW + lane;
            if (nav == 0u) nx0 = L.inr[at]; else nx1 = L.inr[at];
            ++nav; ++in_r;
        }
    };
    auto take = [&](uint32_t n) { const uint32_t v = (uint32_t)bb & ((1u << n) - 1u); bb >>= n; nb -= n; return v; };
    auto push_entry = [&](uint32_t e) { L.entr[(ent_w & (ENT_RING - 1)) * IW + lane] = e; ++ent_w; };
    auto fail = [&](uint32_t why) { bad = why; state = ST_FIN; };
    auto literal = [&](uint32_t byte) {
        lit_acc |= byte << (8u * (lit_n & 3u));
        ++lit_n;
        if ((lit_n & 3u) == 0u) { L.outr[(((lit_n >> 2) - 1u) & (OUT_RING - 1)) * IW + lane] = lit_acc; lit_acc = 0; }
        ++op;
        if (++run == 255u) { push_entry(255u); run = 0; }
    };
    // What the symbol step leaves aside (it sets want / slow and the lane waits): twice per period, for all waiting lanes together.
    //   want 1 / 2: a literal/length / distance code longer than its root table's bits -- one pass of loads for the limits per length,
    //     one for the symbol; the lane goes on with (wsym, wlen).  Any lane's long code would otherwise cost every lane of the wave
    //     the slow path, in most steps;
    //   slow 1: a symbol that is no literal, length or distance: the end of the block, or outside the alphabet; 2: text beyond ISIZE;
    //   3: the 255th literal of a run (an entry of length 0 carries them); 4: a match that reaches outside the text;
    //   and the bytes of a stored block, four at a time.
    uint32_t want = 0, slow = 0, have_w = 0, wsym = 0, wlen = 0, phase = 0, pend_len = 0;
    auto rare_cases = [&]() {
        if (__ballot((want | slow) != 0u || state == ST_STORED) == 0ull) return;
        if (want != 0u) {
            const bool is_lit = want == 1u;
            const uint32_t* const wk = is_lit ? L.lwalk : L.dwalk;
            const uint32_t tb = is_lit ? (uint32_t)LT : (uint32_t)DT;
            const uint32_t v = bitrev((uint32_t)bb & 0x7FFFu, 15);
            uint32_t w[WALK_N];
#pragma unroll
            for (int i = 0; i < WALK_N; ++i) w[i] = wk[i * IW + lane];
            uint32_t len = 0, idx = 0;
#pragma unroll
            for (int i = WALK_N - 1; i >= 0; --i) // (rows past length 15 hold limit 0; the shift stays in range for them)
                if (v < (w[i] & 0xFFFFu)) { len = tb + 1u + (uint32_t)i; idx = ((w[i] >> 16) + (v >> ((14u - tb - (uint32_t)i) & 15u))) & 0xFFFFu; }
            if (len == 0u || idx >= (is_lit ? (uint32_t)LONG_CAP : (uint32_t)DLONG_CAP)) fail(13);
            else { wsym = (is_lit ? L.lsort : L.dsort)[idx * IW + lane]; wlen = len; have_w = 1; }
            want = 0;
        }
        if (slow != 0u) {
            if (slow == 1u) {
                if (phase == 0u && wsym == 256u) { take(wlen); state = final_block ? ST_DONE : ST_BLOCK; }
                else fail(14);
                have_w = 0;
            } else if (slow == 2u) fail(16);
            else if (slow == 3u) { push_entry(255u); run = 0; }
            else fail(20);
            slow = 0;
        }
        if (__ballot(state == ST_STORED) != 0ull) {
            for (int j = 0; j < 4; ++j) {
                topup(); refill();
                if (state == ST_STORED && nb >= 8u && ent_w - ent_f <= (uint32_t)(ENT_RING - 2) && (lit_n >> 2) - out_f < (uint32_t)(OUT_RING - 1)) {
                    literal(take(8));
                    if (--stored_left == 0u) state = final_block ? ST_DONE : ST_BLOCK;
                }
            }
        }
    };

    bool first_period = true;
    uint32_t periods = 0;
    bool hdr_on = true; // (every stream begins with a header)
#ifdef RK_INFLATE_DEBUG
    uint32_t dbg_steps = 0, dbg_sym = 0, dbg_hdr = 0, dbg_long = 0, dbg_build = 0, dbg_symlanes = 0;
#endif
    for (;;) {
        // (a stream that cannot go on -- its input used up, or nothing decoded for far longer than a member takes -- is the host's)
        if (state != ST_FIN && state != ST_DONE && ((!pending && in_w + 4u > in_lim && in_r == in_w && nav == 0u && nb <= 32u) || periods > (STREAM ? 6000000u : 400000u))) fail(2);
        ++periods;
        if constexpr (STREAM) { // what a period can add at most (8 entries, 2 literal dwords, a carry entry) still fits
            if (state != ST_FIN && ent_w + ((lit_n + 3u) >> 2) + 24u > region_dw) fail(22);
            if (state != ST_FIN && state != ST_DONE && !first_period && 32u * (w0 + in_r - nav) - nb > stop_bit + (1u << 27)) fail(23);
        }
        // ---- the wave's memory traffic: once per period, for all lanes
        if (pending) {
#pragma unroll
            for (int i = 0; i < 4; ++i) L.inr[((in_w + i) & (IN_RING - 1)) * IW + lane] = pend.v[i];
            in_w += 4u;
        }
        if (first_period) { // the stream begins inside its first dword
            first_period = false;
            topup(); topup(); refill(); topup();
            if (live && nb) { const uint32_t skip = (me.in_off & 3u) * 8u + start_skip; bb >>= skip; nb -= skip; }
        }
        if (state == ST_DONE && ent_w - ent_f < (uint32_t)ENT_RING) { // the tail: literals after the last match, the last partial dword
            if (!STREAM && op != me.out_len) fail(21);
            else {
                if constexpr (STREAM) end_bit = 32u * (w0 + in_r - nav) - nb; // (nothing was taken since the block ended)
                push_entry(run);
                if (lit_n & 3u) L.outr[((lit_n >> 2) & (OUT_RING - 1)) * IW + lane] = lit_acc;
                state = ST_FIN;
            }
        }
        out_target = bad ? out_f : (state == ST_FIN ? (lit_n + 3u) >> 2 : lit_n >> 2);
        if (bad) ent_f = ent_w;
#pragma unroll
        for (int f = 0; f < 2; ++f)
            if (out_f < out_target) { *(lits_top - out_f) = L.outr[(out_f & (OUT_RING - 1)) * IW + lane]; ++out_f; }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (__ballot(ent_f < ent_w) == 0ull) break;
            if (ent_f < ent_w) { ents[ent_f] = L.entr[(ent_f & (ENT_RING - 1)) * IW + lane]; ++ent_f; }
        }
        pending = state != ST_FIN && in_w - in_r <= (uint32_t)(IN_RING - 4) && in_w + 4u <= in_lim;
        if (pending) pend = *reinterpret_cast<const Quad*>(src + in_w);
        if (__ballot(state != ST_FIN || out_f < out_target || ent_f < ent_w) == 0ull) break;

        // ---- block headers: the lanes that are at one.  Header steps cost the wave the same whether one lane or forty take them, and
        // the lanes reach their blocks' ends at different times: a lane at a header waits for the next window (every HDR_EVERY
        // periods, or HDR_LANES lanes waiting, or nothing else left to do), and a window lasts until no lane is at a header
        {
            const uint64_t at_hdr = __ballot(state < ST_WAIT);
            if (!hdr_on && at_hdr && ((periods & (HDR_EVERY - 1)) == 0u || __popcll(at_hdr) >= HDR_LANES || __ballot(state >= ST_SYM && state < ST_DONE) == 0ull)) hdr_on = true;
            if (!at_hdr) hdr_on = false;
        }
        for (int h = 0; hdr_on && h < K_HDR; ++h) {
            if (__ballot(state < ST_WAIT) == 0ull) break;
#ifdef RK_INFLATE_DEBUG
            ++dbg_hdr;
#endif
            topup(); refill(); topup();
            if constexpr (STREAM) {
                // at a block boundary: the chunk ends here once the stop position is reached (the next chunk begins at this very bit --
                // the host checks that it does; a chunk that runs 16 MB past it without meeting a boundary is no deflate data: above)
                if (state == ST_BLOCK && 32u * (w0 + in_r - nav) - nb >= stop_bit) state = ST_DONE;
            }
            if (state == ST_BLOCK) {
                if (nb >= 3u) {
                    final_block = take(1);
                    const uint32_t type = take(2);
                    if (type == 0u) { take(nb & 7u); state = ST_STORED_HDR; }
                    else if (type == 1u) { nlit = 288; ndist = 30; need_build = 3; state = ST_WAIT; }
                    else if (type == 2u) state = ST_DYN_HDR;
                    else fail(4);
                }
            } else if (state == ST_STORED_HDR) {
                if (nb >= 32u) {
                    const uint32_t len = take(16), nlen = take(16);
                    if ((len ^ 0xFFFFu) != nlen || op + len > me.out_len) fail(3);
                    else { stored_left = len; state = len ? ST_STORED : (final_block ? ST_DONE : ST_BLOCK); }
                }
            } else if (state == ST_DYN_HDR) {
                if (nb >= 14u) {
                    nlit = take(5) + 257u; ndist = take(5) + 1u; ncl = take(4) + 4u;
                    if (nlit > 286u || ndist > 30u) fail(5);
                    else {
                        for (int i = 0; i < 5; ++i) L.lens[(CL_AT / 4 + i) * IW + lane] = 0;
                        hi = 0; state = ST_CL_READ;
                    }
                }
            } else if (state == ST_CL_READ) {
                for (int j = 0; j < 8 && hi < ncl && nb >= 3u; ++j) { set_len(L, CL_AT + L.clorder[hi], lane, take(3)); ++hi; }
                if (hi == ncl) { need_build = 1; state = ST_WAIT; }
            } else if (state == ST_LENS) {
                if (nb >= 14u) {
                    const uint32_t e = L.lit[((uint32_t)bb & ((1u << CT) - 1u)) * IW + lane];
                    const uint32_t l = e >> 12, sym = e & 0xFFFu;
                    if (l == 0u) fail(7);
                    else {
                        take(l);
                        if (sym < 16u) { set_len(L, (int)hi, lane, sym); prev = sym; ++hi; }
                        else {
                            uint32_t rep, val = 0;
                            if (sym == 16u) { rep = 3u + take(2); val = prev; if (hi == 0u) rep = 1000u; }
                            else if (sym == 17u) { rep = 3u + take(3); prev = 0; }
                            else { rep = 11u + take(7); prev = 0; }
                            if (hi + rep > nlit + ndist) fail(9);
                            else { for (uint32_t j = 0; j < rep; ++j) set_len(L, (int)(hi + j), lane, val); hi += rep; }
                        }
                        if (state == ST_LENS && hi == nlit + ndist) {
                            if (len_at(L, 256, lane) == 0u) fail(10);
                            else { need_build = 2; state = ST_WAIT; }
                        }
                    }
                }
            }
        }
        // ---- code tables: the wave builds them, one asking lane at a time
        for (uint64_t req = __ballot(need_build != 0u && state == ST_WAIT); req; req &= req - 1ull) {
            const int t = __builtin_ctzll(req);
#ifdef RK_INFLATE_DEBUG
            ++dbg_build;
#endif
            const uint32_t kind = (uint32_t)__builtin_amdgcn_readlane((int)need_build, t);
            bool ok;
            if (kind == 1u) {
                isync();
                uint32_t lc[1];
                load_lens<1>(L, t, CL_AT, 19, lane, lc);
                ok = coop_build<CT, 1>(lc, t, L.lit, L.dsort, 0, nullptr, lane, true); // (cap 0: a code-length code has no symbol beyond the root's 7 bits)
            } else {
                const int nl = __builtin_amdgcn_readlane((int)nlit, t), nd = __builtin_amdgcn_readlane((int)ndist, t);
                if (kind == 3u) { // the fixed codes (RFC 1951 3.2.6), four symbols per word: 144 x 8, 112 x 9, 24 x 7, 8 x 8 bits; 30 (32) x 5
                    for (int i = lane; i < 80; i += IW) L.lens[i * IW + t] = (uint16_t)(i < 36 ? 0x8888 : (i < 64 ? 0x9999 : (i < 70 ? 0x7777 : (i < 72 ? 0x8888 : 0x5555))));
                }
                isync();
                uint32_t ll[5], ld[1];
                load_lens<5>(L, t, 0, nl, lane, ll);
                load_lens<1>(L, t, nl, nd, lane, ld);
                isync(); // (every length is in a register before the sorted list takes their rows)
                ok = coop_build<LT, 5>(ll, t, L.lit, L.lsort, LONG_CAP, L.lwalk, lane, false);
                // (the fixed distance code, 30 codes of 5 bits, is incomplete by definition: RFC 1951 3.2.6)
                ok = coop_build<DT, 1>(ld, t, L.dist, L.dsort, DLONG_CAP, L.dwalk, lane, false, kind == 3u) && ok;
            }
            if (lane == t) {
                need_build = 0;
                if (!ok) fail(11);
                else { state = kind == 1u ? ST_LENS : ST_SYM; hi = 0; prev = 0; }
            }
        }
        // ---- symbols: one Huffman symbol per step and lane -- a literal, a length, or (the step after a length) its distance.
        // The step is straight-line code: every lane pays for every branch any lane takes, so the three common cases are computed
        // side by side and selected, ring stores that are not due go to a junk slot, and whatever is rare -- an end of block, a code
        // longer than the root's bits, the 255th literal of a run, a stored block's bytes, every error -- is left to
        // rare_cases(), twice per period, while the lane waits.
        for (int it = 0; it < K_SYM; ++it) {
            topup();
            {
                const bool c = nb <= 32u && nav != 0u;
                bb |= c ? (uint64_t)nx0 << (nb & 63u) : 0ull;
                nb += c ? 32u : 0u; nx0 = c ? nx1 : nx0; nav -= c ? 1u : 0u;
            }
            const bool room = ent_w - ent_f <= (uint32_t)(ENT_RING - 2) && (lit_n >> 2) - out_f < (uint32_t)(OUT_RING - 1); // (the partial dword's slot stays free)
            const bool act = state == ST_SYM && nb > 32u && room && (want | slow) == 0u;
            const bool dstep = phase != 0u;
            const uint32_t slot = dstep ? (uint32_t)offsetof(Lds, dist) / 2u + ((uint32_t)bb & ((1u << DT) - 1u)) * IW
                                        : (uint32_t)offsetof(Lds, lit) / 2u + ((uint32_t)bb & ((1u << LT) - 1u)) * IW;
            const uint32_t e = reinterpret_cast<const uint16_t*>(&L)[slot + (uint32_t)lane];
            const uint32_t l = have_w ? wlen : e >> 12, sym = have_w ? wsym : e & 0xFFFu;
            const bool coded = l != 0u;
            const bool is_lit = !dstep && sym < 256u, is_len = !dstep && sym - 257u < 29u, is_dist = dstep && sym < 30u;
            const bool known = act && coded && (is_lit || is_len || is_dist);
            const bool go = known && !(is_lit && op >= me.out_len);
            want = act && !coded ? (dstep ? 2u : 1u) : want;                 // a long code, or no code at all: rare_cases() finds out
            if (act && coded && !go) { slow = known ? 2u : 1u; have_w = 1; wlen = l; wsym = sym; } // end of block / a symbol outside the alphabet / text beyond ISIZE
            have_w = go ? 0u : have_w;
            // base and extra bits of a length (RFC 1951 3.2.5: four codes per extra bit) and of a distance (two per extra bit)
            const uint32_t li = sym - 257u;
            const uint32_t leb = li < 8u || li == 28u ? 0u : (li - 4u) >> 2;
            const uint32_t lbase = li < 8u ? li + 3u : (li == 28u ? 258u : 3u + ((4u + (li & 3u)) << (leb & 7u)));
            const uint32_t deb = sym < 4u ? 0u : (sym - 2u) >> 1;
            const uint32_t dbase = sym < 4u ? sym + 1u : 1u + ((2u + (sym & 1u)) << (deb & 15u));
            const uint32_t eb = !go || is_lit ? 0u : (is_len ? leb : deb) & 15u;
            const uint32_t n1 = go ? l : 0u;
            bb >>= n1;
            const uint32_t x = (uint32_t)bb & ((1u << eb) - 1u);
            bb >>= eb;
            nb -= n1 + eb;
            // a literal
            const bool lit = go && is_lit;
            lit_acc |= lit ? sym << (8u * (lit_n & 3u)) : 0u;
            const bool full = lit && (lit_n & 3u) == 3u;
            L.outr[full ? ((lit_n >> 2) & (OUT_RING - 1)) * IW + lane : OUT_RING * IW + lane] = lit_acc; // (row OUT_RING: the junk row)
            lit_acc = full ? 0u : lit_acc;
            lit_n += lit ? 1u : 0u; op += lit ? 1u : 0u; run += lit ? 1u : 0u;
            slow = lit && run == 255u ? 3u : slow;
            // a length: its distance is the next step's
            pend_len = go && is_len ? lbase + x : pend_len;
            phase = go && is_len ? 1u : phase;
            // a distance: the match becomes an entry
            const uint32_t dist = dbase + x;
            const bool dgo = go && is_dist;
            const bool match = dgo && (dist <= op || (STREAM && !no_text_before)) && op + pend_len <= me.out_len;
            slow = dgo && !match ? 4u : slow;
            L.entr[match ? (ent_w & (ENT_RING - 1)) * IW + lane : ENT_RING * IW + lane] = run | (pend_len << 8) | ((dist - 1u) << 17);
            ent_w += match ? 1u : 0u;
            run = match ? 0u : run;
            op += match ? pend_len : 0u;
            phase = match ? 0u : phase;
#ifdef RK_INFLATE_DEBUG
            ++dbg_steps; dbg_sym += (uint32_t)__popcll(__ballot(go)); dbg_symlanes += (uint32_t)__popcll(__ballot(state == ST_SYM));
            if ((it & 3) == 3 && __ballot((want | slow) != 0u)) ++dbg_long;
#endif
            if ((it & 3) == 3) rare_cases();
        }
    }
#ifdef RK_INFLATE_DEBUG
    if (blockIdx.x == 0 && lane == 0) printf("inflate dbg: periods %u steps %u symbols %u (%.1f lanes per step; %.1f lanes in SYM state) header iterations %u long-code passes %u builds %u\n", periods, dbg_steps, dbg_sym, (double)dbg_sym / dbg_steps, (double)dbg_symlanes / dbg_steps, dbg_hdr, dbg_long, dbg_build);
#endif
    if constexpr (STREAM) {
        if (live) {
            GzChunk& g = reinterpret_cast<GzChunk*>(const_cast<void*>(mem_))[m];
            g.status = bad; g.end_bit = end_bit; g.final_seen = bad ? 0u : final_block;
            g.out_len = bad ? 0u : op; g.nent = bad ? 0u : ent_w; g.nlit = bad ? 0u : lit_n;
        }
    } else {
        if (live) { status[m] = bad; status[nmem + m] = bad ? 0u : ent_w | (lit_n << 15); } // (at most 22 106 entries, 65 536 literals)
    }
}

// Pass 2: the member's text built in LDS from its literal stream and its entries, and written out as it becomes final.
// Round 5's form took one batch of 64 entries at a time straight from global memory -- entries, then (their place known) literals: two
// dependent round trips per batch, ~5 us each -- in a 64 KB window, two members per CU: 1 ms per member, as long as pass 1.  Now:
//   * the inputs come in bulk: 512 entries and 1.5 KB of the literal stream are staged in LDS together (every load in flight at once);
//   * one packed prefix sum (DPP) places every literal run and match of a batch; a lane copies its own run (the first 8 bytes, from
//     one 8-byte read of the stage; longer runs are finished by the whole wave); matches resolve in ROUNDS behind a frontier:
//     F = the first unresolved match -- everything below its start is final, so every match whose source ends there (and F itself,
//     whose source may overlap its own output) is copied now, a lane each (three 8-byte reads, then stores that wait for nothing),
//     long ones by the whole wave.  Level-1 FASTQ takes ~7 rounds per batch;
//   * the window is 35 KB, not 64: a match reaches at most 32 KB back, so once a batch is final its 16-byte groups leave for global
//     memory and window index k >= P2_R reuses the place of k - P2_R (a batch is cut so that it spans < P2_R - 32 KB).  39.5 KB of
//     LDS per member: FOUR members per CU (or two beside a pass-1 wave) instead of two -- the kernel is bound by the latency of its
//     dependent LDS steps, not by any throughput, so members in flight are what it scales with.
// The window is addressed like the text (window index k <-> address of the text - its low four bits + k), so it leaves through
// aligned 16-byte LDS reads and global stores.
struct __attribute__((packed, aligned(1))) lds_u64 { uint64_t v; }; // eight bytes at any LDS address (gfx950 runs with unaligned DS access)
constexpr int P2_SB = 512;                 // entries staged at a time
constexpr int P2_LIT = 1536;               // literal bytes staged at a time (+ 4: the stage begins at a dword boundary)
constexpr uint32_t P2_R = 35840;           // window bytes (a multiple of 16): index k and k + P2_R share a place
constexpr uint32_t P2_SPAN = 2816;         // most bytes a batch may produce: P2_R > 32768 + P2_SPAN + 16
static_assert(P2_R % 16 == 0 && P2_R > 32768 + P2_SPAN + 16 && P2_SPAN >= 513, "a batch's sources must outlive the batch; one entry always fits");
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1 (a lane without a source adds 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    return v;
}
// STREAM (chunks of an ordinary gzip stream, rk_gunzip.hip): a chunk's matches may copy from the 32 KB of text in front of it, which
// nobody knows yet when the chunks are placed side by side.  The window is then given 32 KB of PATTERN in front of the text --
// window index j < 32768 holds pattern(j) -- and the chunk is placed three times, into three planes: pattern A (j & 255),
// B (j >> 8), C (255).  A byte of the text that is its own shows the same value in all planes; one that was copied (through any chain
// of matches) from byte j of the text in front shows (j & 255, j >> 8, 255), and j >> 8 < 128 tells the two apart when all else is
// equal -- k_gz_resolve turns the planes into text once the windows are known (k_gz_windows, in stream order).
// A chunk may be of any length: window indices wrap modulo P2_R as often as it takes.
template <bool STREAM>
__global__ __launch_bounds__(IW) void k_inflate_place(const void* __restrict__ mem_, uint32_t nmem, uint8_t* __restrict__ text, const uint32_t* __restrict__ scratch,
                                                      const uint32_t* __restrict__ status, uint32_t group, size_t plane_stride) {
    __shared__ __attribute__((aligned(16))) uint8_t win[P2_R + 16];
    __shared__ uint32_t s_ent[P2_SB];
    __shared__ __attribute__((aligned(16))) uint32_t s_lit[P2_LIT / 4 + 4];
    const int lane = threadIdx.x;
    // STREAM: a workgroup places `group` consecutive chunks as ONE stretch of text in one window -- only the first of them needs the
    // stand-in pattern, and the window walk (k_gz_windows, the one sequential kernel of the route) has a step per group, not per
    // chunk; blockIdx.y = the plane (its pattern, its copy of the text).
    const uint32_t m = STREAM ? blockIdx.x * group : blockIdx.x;
    const uint32_t pattern = STREAM ? blockIdx.y : 0u;
    if constexpr (STREAM) text += (size_t)pattern * plane_stride;
    if (m >= nmem) return;
    uint32_t nent, nlit, n, out_off, match_off, region_dw = 0, nsub = 1, n_end;
    if constexpr (STREAM) {
        const GzChunk* const gc = reinterpret_cast<const GzChunk*>(mem_);
        const GzChunk g = gc[m];
        nent = g.nent; nlit = g.nlit; n = g.out_len; out_off = g.out_off; match_off = g.scratch_off; region_dw = g.scratch_dw;
        n_end = n;
        nsub = nmem - m < group ? nmem - m : group;
        for (uint32_t j = 1; j < nsub; ++j) n += gc[m + j].out_len;
    } else {
        if (status[m] != 0u) return;
        nent = status[nmem + m] & 0x7FFFu; nlit = status[nmem + m] >> 15;
        const InflateMember me = reinterpret_cast<const InflateMember*>(mem_)[m];
        n = me.out_len; out_off = me.out_off; match_off = me.match_off;
        n_end = n;
    }
    constexpr uint32_t K0 = STREAM ? 32768u : 0u; // window index of the first 16-byte group that holds text
    const uint32_t* ents = scratch + match_off;
    const uint32_t* lits_top = ents + ((STREAM ? region_dw : scratch_dwords(n)) - 1u); // literal dword d lives at lits_top[-d] (pass 1)
    auto lit_global = [&](uint32_t idx) -> uint8_t { return (uint8_t)(*(lits_top - (idx >> 2)) >> (8u * (idx & 3u))); };
    uint8_t* const dst = text + out_off;
    const uint32_t wb = K0 + (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u); // window index of the text's first byte
    uint8_t* const a0 = dst - wb;                                            // address of window index 0 (16-byte aligned)
    const uint8_t* const sl8 = reinterpret_cast<const uint8_t*>(s_lit);
    // place of window index k (members: k < 2 P2_R, a member is at most 64 KB)
    auto wi = [](uint32_t k) -> uint32_t { if constexpr (STREAM) return k % P2_R; else return k >= P2_R ? k - P2_R : k; };
    uint32_t kp = wb, lp0 = 0;           // window index of the next output byte; position in the literal stream
    uint32_t flushed = K0;               // window indices below it are in global memory (a multiple of 16, or the head's end)
    if constexpr (STREAM) {
        // (the text's first byte has window index wb: byte j of the window in front of it -- j = 0: 32 KB back -- has index wb - 32768 + j)
        for (uint32_t j = (uint32_t)lane; j < 32768u; j += IW) win[wb - K0 + j] = (uint8_t)(pattern == 0u ? j : (pattern == 1u ? j >> 8 : 255u));
        isync();
    }
    uint32_t lit_base = 0, lit_have = 0; // s_lit holds the literal bytes [lit_base, lit_base + lit_have)
    constexpr uint32_t LIT_DW = P2_LIT / 4 + 1, LIT_K = (LIT_DW + IW - 1) / IW;
    uint32_t lit_dw_end = (nlit + 3u) >> 2; // dwords pass 1 wrote
    auto stage_lits = [&]() { // the literal stream from where it stands (wave-uniform call)
        uint32_t lv[LIT_K];
        lit_base = lp0 & ~3u;
#pragma unroll
        for (uint32_t k = 0; k < LIT_K; ++k) { const uint32_t i = (uint32_t)lane + IW * k, d = (lit_base >> 2) + i; lv[k] = (i < LIT_DW && d < lit_dw_end) ? *(lits_top - d) : 0u; }
        isync(); // (earlier batches have read their literals)
#pragma unroll
        for (uint32_t k = 0; k < LIT_K; ++k) { const uint32_t i = (uint32_t)lane + IW * k; if (i < LIT_DW) s_lit[i] = lv[k]; }
        lit_have = nlit - lit_base < LIT_DW * 4u ? nlit - lit_base : LIT_DW * 4u;
        isync();
    };
    // whole 16-byte groups below window index `upto` leave (what is below the current batch is final)
    auto flush_to = [&](uint32_t upto) {
        if (flushed == K0 && wb != K0 && upto >= K0 + 16u) { // the first group holds bytes in front of the text: its text bytes one by one
            if ((uint32_t)lane >= wb - K0 && lane < 16) a0[K0 + lane] = win[K0 + lane];
            flushed = K0 + 16u;
        }
        const uint32_t hi = upto & ~15u;
        for (uint32_t k = flushed + 16u * (uint32_t)lane; k < hi; k += 16u * IW) *reinterpret_cast<uint4*>(a0 + k) = *reinterpret_cast<const uint4*>(win + wi(k));
        if (hi > flushed) flushed = hi;
    };
    for (uint32_t sub = 0; sub < nsub; ++sub) {
    if constexpr (STREAM) {
        if (sub) { // the next chunk of the group: its entries and literals, the same window and text
            const GzChunk g = reinterpret_cast<const GzChunk*>(mem_)[m + sub];
            nent = g.nent; nlit = g.nlit; n_end += g.out_len;
            ents = scratch + g.scratch_off; lits_top = ents + (g.scratch_dw - 1u);
            lit_dw_end = (nlit + 3u) >> 2; lp0 = 0; lit_base = 0; lit_have = 0;
        }
    }
    for (uint32_t e0 = 0; e0 < nent; e0 += P2_SB) {
        const uint32_t ne = nent - e0 < (uint32_t)P2_SB ? nent - e0 : (uint32_t)P2_SB;
        { // stage the entries of this stretch, and the literals again from where the stream stands
            uint32_t ev[P2_SB / IW];
#pragma unroll
            for (int k = 0; k < P2_SB / IW; ++k) { const uint32_t i = (uint32_t)lane + IW * k; ev[k] = i < ne ? ents[e0 + i] : 0u; }
            isync();
#pragma unroll
            for (int k = 0; k < P2_SB / IW; ++k) s_ent[lane + IW * k] = ev[k];
            stage_lits();
        }
        for (uint32_t q0 = 0; q0 < ne;) {
            const uint32_t e = q0 + (uint32_t)lane < ne ? s_ent[q0 + lane] : 0u;
            uint32_t run = e & 255u, len = (e >> 8) & 511u;
            const uint32_t dist = (e >> 17) + 1u;
            // inclusive sums over the lanes, both in one word: literals (< 2^14) | literals + match bytes (< 2^16) << 16
            uint32_t sc = wave_incl_scan_add(run | ((run + len) << 16));
            // the batch = the leading entries that produce at most P2_SPAN bytes together (the first one always does)
            const uint32_t take = (uint32_t)__popcll(__ballot((sc >> 16) <= P2_SPAN));
            if ((uint32_t)lane >= take) { run = 0; len = 0; }
            const uint32_t r = sc & 0xFFFFu, t = sc >> 16;
            const uint32_t tot_r = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)take - 1), tot_t = (uint32_t)__builtin_amdgcn_readlane((int)t, (int)take - 1);
            if (kp - wb + tot_t > n_end || lp0 + tot_r > nlit) return; // (pass 1 checked every entry against out_len: unreachable for its output)
            const uint32_t at = kp + t - len;      // window index where this lane's match begins; its run ends there (lanes past the batch: unused)
            const uint32_t lsrc = lp0 + r - run;   // its run's first byte in the literal stream
            // the batch's literals: from the stage, which moves up when the batch reaches past it (a batch with more literals than
            // the stage holds -- an all-literal stretch -- reads them from global memory where they lie)
            bool direct = false;
            if (lp0 + tot_r > lit_base + lit_have) { // wave-uniform
                if (tot_r + 3u <= (uint32_t)P2_LIT) stage_lits(); else direct = true;
            }
            auto lit = [&](uint32_t idx) -> uint8_t { return direct ? lit_global(idx) : sl8[idx - lit_base]; };
            // literal runs: the first 8 bytes by the run's own lane (one 8-byte read of the stage, byte stores that wait for nothing) ...
            {
                const uint32_t o = at - run;
                if (direct) {
#pragma unroll
                    for (uint32_t b = 0; b < 8u; ++b) if (b < run) win[wi(o + b)] = lit_global(lsrc + b);
                } else {
                    const uint64_t v = reinterpret_cast<const lds_u64*>(sl8 + (run ? lsrc - lit_base : 0u))->v; // (the stage is followed by 12 spare bytes)
#pragma unroll
                    for (uint32_t b = 0; b < 8u; ++b) if (b < run) win[wi(o + b)] = (uint8_t)(v >> (8u * b));
                }
            }
            // ... the rest of a longer run by the whole wave
            for (uint64_t lm = __ballot(run > 8u); lm; lm &= lm - 1ull) {
                const int j = __builtin_ctzll(lm);
                const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)run, j), oj = (uint32_t)__builtin_amdgcn_readlane((int)at, j) - rj,
                               sj = (uint32_t)__builtin_amdgcn_readlane((int)lsrc, j);
                for (uint32_t i = 8u + (uint32_t)lane; i < rj; i += IW) win[wi(oj + i)] = lit(sj + i);
            }
            isync();
            // matches, in rounds behind the frontier
            for (uint64_t um = __ballot(len != 0u); um;) {
                const int F = __builtin_ctzll(um);
                const uint32_t atF = (uint32_t)__builtin_amdgcn_readlane((int)at, F);
                const bool mine = ((um >> lane) & 1ull) != 0ull && (at - dist + len <= atF || lane == F);
                const uint32_t ks = at - dist; // window index of the source
                // a copy that would touch both sides of the wrap at P2_R (source or destination; one match in a thousand) goes byte by byte
                const bool wrap = STREAM ? (wi(ks) + 24u > P2_R || wi(at) + len > P2_R) : ((ks < P2_R && ks + 24u > P2_R) || (at < P2_R && at + len > P2_R));
                for (uint64_t lm = __ballot(mine && len >= 24u); lm; lm &= lm - 1ull) { // long: 64 lanes per copy; a source that overlaps its output repeats with period dist
                    const int j = __builtin_ctzll(lm);
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)at, j), ln = (uint32_t)__builtin_amdgcn_readlane((int)len, j),
                                   ds = (uint32_t)__builtin_amdgcn_readlane((int)dist, j);
                    if (ds >= ln) { for (uint32_t i = (uint32_t)lane; i < ln; i += IW) win[wi(o + i)] = win[wi(o - ds + i)]; }
                    else { for (uint32_t i = (uint32_t)lane; i < ln; i += IW) win[wi(o + i)] = win[wi(o - ds + i % ds)]; }
                }
                // short and apart from its source (all of them but, at most, F): the source in three 8-byte reads, then stores that
                // wait for nothing -- a byte-by-byte copy pays the LDS round trip twice per byte
                const bool sh = mine && len < 24u && dist >= len && !wrap;
                {
                    const uint8_t* const sp = win + (sh ? wi(ks) : 0u);
                    uint64_t r0 = reinterpret_cast<const lds_u64*>(sp)->v;
                    const uint64_t r1 = reinterpret_cast<const lds_u64*>(sp + 8)->v, r2 = reinterpret_cast<const lds_u64*>(sp + 16)->v;
                    if (sh) {
                        uint8_t* dp = win + wi(at);
                        uint32_t left = len;
                        if (left >= 8u) { reinterpret_cast<lds_u64*>(dp)->v = r0; dp += 8; left -= 8u; r0 = r1; }
                        if (left >= 8u) { reinterpret_cast<lds_u64*>(dp)->v = r0; dp += 8; left -= 8u; r0 = r2; }
#pragma unroll
                        for (uint32_t b = 0; b < 7u; ++b) if (b < left) dp[b] = (uint8_t)(r0 >> (8u * b));
                    }
                }
                // short and overlapping its own output (F only: a repeat with a period below its length), or across the wrap: byte after byte
                const bool ov = mine && len < 24u && (dist < len || wrap);
                if (__ballot(ov) != 0ull) { if (ov) for (uint32_t i = 0; i < len; ++i) win[wi(at + i)] = win[wi(ks + i)]; }
                um &= ~__ballot(mine);
                isync();
            }
            kp += tot_t; lp0 += tot_r; q0 += take;
            flush_to(kp);
        }
    }
    }
    isync();
    // what is left: the head (a text that ends inside the first group), the last whole groups, the tail bytes
    const uint32_t endk = wb + n;
    if (flushed == K0 && wb != K0) {
        const uint32_t hb = endk < K0 + 16u ? endk : K0 + 16u;
        if (K0 + (uint32_t)lane >= wb && K0 + (uint32_t)lane < hb) a0[K0 + lane] = win[K0 + lane];
        flushed = hb;
    }
    if (flushed < endk) {
        flush_to(endk);
        if (flushed + (uint32_t)lane < endk) a0[flushed + lane] = win[wi(flushed + lane)];
    }
}

// The members' CRC-32 (gzread checks it for every member the reference reads, /root/reference/src/rkmh.cpp:238-263): one wave per
// member over the text pass 2 wrote, the lanes' pieces and the recombination as rk_crc32.hpp states them (pinned against zlib on
// the host by tools/crc32_check.cpp).  A member whose text does not give the CRC-32 of its footer gets status 30 -- the caller hands
// the job to the host inflater, which reports the damage.
__device__ const Crc32Tables CRC32_TABLES = make_crc32_tables();
// the wave's CRC-32 of text[off, off + len), len <= 65536 (every lane returns it); byte_t = the byte table in LDS
__device__ __forceinline__ uint32_t wave_crc32(const uint32_t* byte_t, const uint8_t* __restrict__ text, uint32_t off, uint32_t len, uint32_t lane) {
    const Crc32Piece p = crc32_piece(off, len, lane);
    uint32_t s = p.init;
    for (uint32_t g = p.b & ~15u; g < p.e; g += 16u) { // (the text buffer is 16-byte aligned and readable a little past its end)
        const uint4 v = *reinterpret_cast<const uint4*>(text + g);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t at = g + (uint32_t)j, byte = (w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const uint32_t nxt = byte_t[(s ^ byte) & 0xFFu] ^ (s >> 8);
            s = (at >= p.b && at < p.e) ? nxt : s;
        }
    }
    uint32_t x = (p.e > p.b || lane == 0u) ? crc32_advance(CRC32_TABLES, s, p.z) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d);
    return ~x;
}
__global__ __launch_bounds__(256) void k_crc32_members(const InflateMember* __restrict__ mem, uint32_t nmem, const uint8_t* __restrict__ text,
                                                       const uint8_t* __restrict__ comp, uint32_t* __restrict__ status) {
    __shared__ uint32_t byte_t[256];
    byte_t[threadIdx.x] = CRC32_TABLES.byte[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, m = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (m >= nmem || status[m] != 0u) return;
    const InflateMember me = mem[m];
    const uint32_t crc = wave_crc32(byte_t, text, me.out_off, me.out_len, lane);
    if (lane == 0u) {
        const uint8_t* f = comp + me.in_off + me.in_len; // the member's footer: CRC-32, ISIZE (little endian, any alignment)
        const uint32_t want = (uint32_t)f[0] | ((uint32_t)f[1] << 8) | ((uint32_t)f[2] << 16) | ((uint32_t)f[3] << 24);
        if (crc != want) status[m] = 30u;
    }
}
// crc[i] = CRC-32 of text[64 K i, 64 K (i + 1)) within [0, n): the host joins them (crc32_combine) into the stream's
__global__ __launch_bounds__(256) void k_crc32_segments(const uint8_t* __restrict__ text, uint32_t n, uint32_t* __restrict__ crc) {
    __shared__ uint32_t byte_t[256];
    byte_t[threadIdx.x] = CRC32_TABLES.byte[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, m = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t off = m << 16;
    if (off >= n) return;
    const uint32_t c = wave_crc32(byte_t, text, off, n - off < 65536u ? n - off : 65536u, lane);
    if (lane == 0u) crc[m] = c;
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

// ---- an ordinary gzip file: ONE deflate stream (rk_gunzip.hip drives these; the reference reads it with gzread, rkmh.cpp:238-263) ----
// Pass 1's output needs no window, so a lane can decode from any block header of the stream.  Block headers are not marked in a
// deflate stream; k_gz_find_starts looks for them: from every chunk boundary (a fixed stride of compressed bytes) on, every bit
// position is tried as the start of a non-final dynamic-Huffman block -- 3 header bits, counts in range, a complete code-length
// code, nlit + ndist code lengths that decode and form complete literal/length and distance codes, an end-of-block code.  Random
// bits pass all of that about once in 10^8..10^9 positions; a false start costs only the work of one lane: the chunk in front of
// it does not end at it (pass 1 reports where every chunk ended), and the host joins the chain from true starts only.
struct __attribute__((packed, aligned(1))) g_u64 { uint64_t v; }; // eight bytes at any global address
struct __attribute__((packed, aligned(1))) g_u128 { uint4 v; }; // sixteen
__device__ __forceinline__ uint64_t gz_peek(const uint8_t* __restrict__ comp, uint32_t pos) { return reinterpret_cast<const g_u64*>(comp + (pos >> 3))->v >> (pos & 7u); } // >= 57 bits
// stages 1 and 2, a few dozen instructions: the three header bits, the counts, a complete code-length code
__device__ __forceinline__ bool gz_header_maybe(const uint8_t* __restrict__ comp, uint32_t pos, uint32_t nbits) {
    if (pos + 4096u > nbits) return false; // (a last block this close to the end is left to the chunk in front of it)
    const uint64_t w = gz_peek(comp, pos);
    if ((w & 7u) != 4u) return false; // BFINAL = 0, BTYPE = 2
    const uint32_t nlit = (uint32_t)((w >> 3) & 31u) + 257u, ndist = (uint32_t)((w >> 8) & 31u) + 1u, ncl = (uint32_t)((w >> 13) & 15u) + 4u;
    if (nlit > 286u || ndist > 30u) return false;
    const uint64_t c = gz_peek(comp, pos + 17u); // (>= 57 bits: all 19 lengths of 3 bits)
    uint32_t kraft = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19u; ++i) {
        const uint32_t l = (uint32_t)(c >> (3u * i)) & 7u;
        kraft += (i < ncl && l) ? 128u >> l : 0u;
    }
    return kraft == 128u; // (zlib refuses an incomplete code-length code: inftrees.c, type CODES)
}
// stage 3: the nlit + ndist code lengths decode and form complete codes.  Everything lives in registers -- the code-length code's
// symbols in (length, symbol) order, 5 bits each, in two words; a first form kept them in arrays, which the compiler put into
// scratch memory: ~1 us per symbol, 300 us per candidate, and one candidate in every other step of 256 positions: 42 ms per stretch
__device__ bool gz_header_ok(const uint8_t* __restrict__ comp, uint32_t pos, uint32_t nbits) {
    uint64_t w = gz_peek(comp, pos);
    const uint32_t nlit = (uint32_t)((w >> 3) & 31u) + 257u, ndist = (uint32_t)((w >> 8) & 31u) + 1u, ncl = (uint32_t)((w >> 13) & 15u) + 4u;
    uint32_t p = pos + 17u;
    const uint64_t c = gz_peek(comp, p);
    p += 3u * ncl;
    // lengths by symbol: cl3 holds 3 bits per symbol 0 .. 18; counts per length in cnt (4 bits each: at most 19... packed 5 bits)
    uint64_t cl3 = 0;
    uint32_t cnt[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) cnt[l] = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19u; ++i) {
        const uint32_t l = i < ncl ? (uint32_t)(c >> (3u * i)) & 7u : 0u;
        cl3 |= (uint64_t)l << (3u * CL_ORDER_C[i]);
#pragma unroll
        for (int k = 1; k < 8; ++k) cnt[k] += l == (uint32_t)k ? 1u : 0u;
    }
    // symbols sorted by (length, symbol): slot k of (s_lo, s_hi), 5 bits each, 12 per word
    uint64_t s_lo = 0, s_hi = 0;
    {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t l = 1; l < 8u; ++l)
#pragma unroll
            for (uint32_t sy = 0; sy < 19u; ++sy) {
                const bool is = ((uint32_t)(cl3 >> (3u * sy)) & 7u) == l;
                if (is) { if (k < 12u) s_lo |= (uint64_t)sy << (5u * k); else s_hi |= (uint64_t)sy << (5u * (k - 12u)); ++k; }
            }
    }
    uint32_t lit_kraft = 0, dist_kraft = 0, lit_max = 0, dist_max = 0, prev = 0, eob_len = 0, have = 0;
    const uint32_t total = nlit + ndist;
    for (uint32_t i = 0; i < total;) {
        if (have < 14u) { w = gz_peek(comp, p); have = 57u; }
        int code = 0, first = 0, index = 0;
        uint32_t s = 99, used = 0;
#pragma unroll
        for (uint32_t l = 1; l <= 7u; ++l) {
            code |= (int)((w >> (l - 1u)) & 1u);
            const int cn = (int)cnt[l];
            if (s == 99u && code - cn < first) {
                const uint32_t k = (uint32_t)(index + (code - first));
                s = k < 12u ? (uint32_t)(s_lo >> (5u * k)) & 31u : (uint32_t)(s_hi >> (5u * (k - 12u))) & 31u;
                used = l;
            }
            index += cn; first += cn; first <<= 1; code <<= 1;
        }
        if (s == 99u) return false;
        w >>= used; p += used; have -= used;
        uint32_t rep = 1, val = s;
        if (s == 16u) { if (i == 0u) return false; rep = 3u + (uint32_t)(w & 3u); w >>= 2; p += 2u; have -= 2u; val = prev; }
        else if (s == 17u) { rep = 3u + (uint32_t)(w & 7u); w >>= 3; p += 3u; have -= 3u; val = 0; }
        else if (s == 18u) { rep = 11u + (uint32_t)(w & 127u); w >>= 7; p += 7u; have -= 7u; val = 0; }
        if (i + rep > total) return false;
        if (s < 16u) prev = s; else if (s != 16u) prev = 0;
        if (val) {
            // the run lies in the literal/length code up to symbol nlit - 1, in the distance code behind it
            const uint32_t in_lit = i < nlit ? (i + rep <= nlit ? rep : nlit - i) : 0u, in_dist = rep - in_lit;
            lit_kraft += in_lit * (32768u >> val); dist_kraft += in_dist * (32768u >> val);
            if (in_lit && val > lit_max) lit_max = val;
            if (in_dist && val > dist_max) dist_max = val;
            if (i <= 256u && i + rep > 256u) eob_len = val;
        }
        i += rep;
        if (p + 64u > nbits) return false;
    }
    if (eob_len == 0u) return false;
    if (lit_kraft > 32768u || (lit_kraft < 32768u && lit_max != 1u)) return false;
    if (dist_kraft > 32768u || (dist_kraft < 32768u && dist_max > 1u)) return false;
    return true;
}
// found[b] = the first bit position in [from[b], to[b]) that looks like a block header, 0xFFFFFFFF if none does.  A workgroup takes the
// range in pieces of 64 K positions: every thread tries its positions with the cheap stages and notes the survivors (one in a few
// hundred) in LDS; then the survivors are checked in full, a thread each; the first piece with a header ends the search.
constexpr uint32_t GZ_PIECE = 65536, GZ_CAND = 4096;
__global__ __launch_bounds__(256) void k_gz_find_starts(const uint8_t* __restrict__ comp, uint32_t nbits, const uint32_t* __restrict__ from, const uint32_t* __restrict__ to,
                                                        uint32_t* __restrict__ found) {
    __shared__ uint32_t best, ncand, cand[GZ_CAND];
    const uint32_t lo = from[blockIdx.x], hi = to[blockIdx.x];
    if (threadIdx.x == 0) best = 0xFFFFFFFFu;
    for (uint32_t base = lo; base < hi; base += GZ_PIECE) {
        if (threadIdx.x == 0) ncand = 0;
        __syncthreads();
        const uint32_t end = hi - base < GZ_PIECE ? hi : base + GZ_PIECE;
        for (uint32_t pos = base + threadIdx.x; pos < end; pos += 256u)
            if (gz_header_maybe(comp, pos, nbits)) { const uint32_t k = atomicAdd(&ncand, 1u); if (k < GZ_CAND) cand[k] = pos; }
        __syncthreads();
        const uint32_t n = ncand < GZ_CAND ? ncand : GZ_CAND; // (more survivors than the list holds -- not deflate data, or all zeros: the rest of them are skipped)
        for (uint32_t k = threadIdx.x; k < n; k += 256u)
            if (gz_header_ok(comp, cand[k], nbits)) atomicMin(&best, cand[k]);
        __syncthreads();
        if (best != 0xFFFFFFFFu) break;
    }
    if (threadIdx.x == 0) found[blockIdx.x] = best;
}

// The windows, in stream order (ONE workgroup: every chunk's window is the one before it plus its own last 32 KB).  rings[c] = the
// 32 KB of text in front of chunk c as a ring -- byte j of that window (j = 0: 32 KB back) lies at rings[c][(heads[c] + j) & 32767];
// rings[0] / heads[0] are given (the text in front of the first chunk; any bytes for the stream's first), rings[1 .. n] are written.
__global__ __launch_bounds__(1024) void k_gz_windows(const GzChunk* __restrict__ chunks, uint32_t nchunk, const uint8_t* __restrict__ planes, size_t plane_stride,
                                                     uint8_t* __restrict__ rings, uint32_t* __restrict__ heads) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[32768 + 16];
    __shared__ uint32_t s_len[256], s_off[256];
    const uint32_t tid = threadIdx.x;
    for (uint32_t r = 0; r < 2u; ++r) reinterpret_cast<uint4*>(ring)[tid + 1024u * r] = reinterpret_cast<const uint4*>(rings)[tid + 1024u * r];
    uint32_t head = heads[0] & 32767u;
    __syncthreads();
    // a thread owns 32 consecutive bytes of the chunk's last 32 KB: two unaligned 16-byte loads per plane, all six in flight together
    // (the first version read them byte by byte: 160 us per chunk -- ten thousand chunks of a 600 MB stretch took 1.6 s)
    for (uint32_t c0 = 0; c0 < nchunk; c0 += 256u) {
        __syncthreads();
        if (tid < 256u && c0 + tid < nchunk) { s_len[tid] = chunks[c0 + tid].out_len; s_off[tid] = chunks[c0 + tid].out_off; }
        __syncthreads();
        const uint32_t cn = nchunk - c0 < 256u ? nchunk - c0 : 256u;
        // the plane words of a chunk do not depend on the ring: they are requested one chunk ahead, so their latency (~2 us from HBM)
        // hides behind the chunk before (the first form waited for them inside every step: 6.6 us per chunk)
        uint32_t w[3][8];
        bool mine = false, whole = false;
        uint32_t take = 0;
        auto fetch = [&](uint32_t ci) {
            const uint32_t n = s_len[ci];
            take = n < 32768u ? n : 32768u;
            const size_t src = (size_t)s_off[ci] + (n - take);
            const uint32_t i0 = tid * 32u;
            mine = i0 < take; whole = i0 + 32u <= take;
            if (whole) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const g_u128* q = reinterpret_cast<const g_u128*>(planes + (size_t)p * plane_stride + src + i0);
                    const uint4 x = q[0].v, y = q[1].v;
                    w[p][0] = x.x; w[p][1] = x.y; w[p][2] = x.z; w[p][3] = x.w; w[p][4] = y.x; w[p][5] = y.y; w[p][6] = y.z; w[p][7] = y.w;
                }
            } else if (mine) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int d = 0; d < 8; ++d) w[p][d] = 0;
                    for (uint32_t b = 0; b < 32u && i0 + b < take; ++b) w[p][b >> 2] |= (uint32_t)planes[(size_t)p * plane_stride + src + i0 + b] << (8u * (b & 3u));
                }
            }
        };
        fetch(0);
        for (uint32_t ci = 0; ci < cn; ++ci) {
            const uint32_t i0 = tid * 32u;
            const bool mine_c = mine, whole_c = whole;
            const uint32_t take_c = take;
            uint32_t wc[3][8];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int d = 0; d < 8; ++d) wc[p][d] = w[p][d];
            if (ci + 1u < cn) fetch(ci + 1u);
            uint32_t o[8];
            if (mine_c) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    // a byte that came out of the window: plane C shows 255 and plane B < 128 (a byte of the text that IS 255 shows 255 in all)
                    const uint32_t cw = wc[2][d], bw = wc[1][d], aw = wc[0][d];
                    uint32_t ow = cw;
                    const uint32_t is255 = cw & (cw >> 1) & (cw >> 2) & (cw >> 3) & (cw >> 4) & (cw >> 5) & (cw >> 6) & (cw >> 7) & 0x01010101u; // bytes of C equal to 255
                    if (is255) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t C = (cw >> (8 * j)) & 255u, B = (bw >> (8 * j)) & 255u, A = (aw >> (8 * j)) & 255u;
                            if (C == 255u && B != 255u) ow = (ow & ~(255u << (8 * j))) | ((uint32_t)ring[(head + (A | (B << 8))) & 32767u] << (8 * j));
                        }
                    }
                    o[d] = ow;
                }
            }
            __syncthreads();
            if (mine_c) {
                const uint32_t at = (head + i0) & 32767u;
                if (whole_c && at + 32u <= 32768u) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) reinterpret_cast<lds_u64*>(ring + at + 8 * d)->v = (uint64_t)o[2 * d] | ((uint64_t)o[2 * d + 1] << 32);
                } else {
                    for (uint32_t b = 0; b < 32u && i0 + b < take_c; ++b) ring[(at + b) & 32767u] = (uint8_t)(o[b >> 2] >> (8u * (b & 3u)));
                }
            }
            head = (head + take_c) & 32767u;
            __syncthreads();
            uint4* const dst = reinterpret_cast<uint4*>(rings + (size_t)(c0 + ci + 1u) * 32768u);
            for (uint32_t r = 0; r < 2u; ++r) dst[tid + 1024u * r] = reinterpret_cast<const uint4*>(ring)[tid + 1024u * r];
            if (tid == 0) heads[c0 + ci + 1u] = head;
        }
    }
}
// text[out_off + i] of every chunk from its planes and the window in front of it; gridDim.y workgroups share a chunk
__global__ __launch_bounds__(256) void k_gz_resolve(const GzChunk* __restrict__ chunks, const uint8_t* __restrict__ planes, size_t plane_stride,
                                                    const uint8_t* __restrict__ rings, const uint32_t* __restrict__ heads, uint8_t* __restrict__ text) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[32768];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = chunks[c].out_len, off = chunks[c].out_off;
    const uint32_t g_lo = off & ~15u, groups = (off + n + 15u - g_lo) >> 4; // whole 16-byte groups (the buffers are 16-byte aligned)
    const uint32_t per = (groups + gridDim.y - 1u) / gridDim.y, ga = per * blockIdx.y, gb = ga + per < groups ? ga + per : groups;
    if (ga >= gb) return;
    for (uint32_t r = tid; r < 2048u; r += 256u) reinterpret_cast<uint4*>(ring)[r] = reinterpret_cast<const uint4*>(rings + (size_t)c * 32768u)[r];
    const uint32_t head = heads[c] & 32767u;
    __syncthreads();
    for (uint32_t g = ga + tid; g < gb; g += 256u) {
        const size_t at = (size_t)g_lo + 16u * (size_t)g;
        const uint4 a = *reinterpret_cast<const uint4*>(planes + at), b = *reinterpret_cast<const uint4*>(planes + plane_stride + at),
                    cc = *reinterpret_cast<const uint4*>(planes + 2 * plane_stride + at);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, cw[4] = {cc.x, cc.y, cc.z, cc.w};
        uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t A = (aw[j >> 2] >> (8 * (j & 3))) & 255u, B = (bw[j >> 2] >> (8 * (j & 3))) & 255u, C = (cw[j >> 2] >> (8 * (j & 3))) & 255u;
            const uint32_t v = (C != 255u || B == 255u) ? C : ring[(head + (A | (B << 8))) & 32767u];
            ow[j >> 2] |= v << (8 * (j & 3));
        }
        if (at >= off && at + 16u <= (size_t)off + n) *reinterpret_cast<uint4*>(text + at) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        else {
            for (int j = 0; j < 16; ++j)
                if (at + j >= off && at + j < (size_t)off + n) text[at + j] = (uint8_t)(ow[j >> 2] >> (8 * (j & 3)));
        }
    }
}

uint32_t inflate_scratch_dwords(uint32_t out_len) { return scratch_dwords(out_len); }
hipError_t launch_inflate_members(const uint8_t* comp, uint32_t comp_bytes, const InflateMember* mem, uint32_t nmem, uint8_t* text, uint32_t* scratch, uint32_t* status,
                                  hipStream_t st) {
    if (!nmem) return hipSuccess;
    // (once per device: the call takes the runtime's lock, and several workers launch from their own threads)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
    if (!((attr_set.load(std::memory_order_acquire) >> dev) & 1ull) || dev == 63) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_lanes<8, 6, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LaneLds<8, 6>));
        if (attr != hipSuccess) return attr;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_inflate_lanes<8, 6, false>), dim3((nmem + IW - 1) / IW), dim3(IW), sizeof(LaneLds<8, 6>), st, comp, comp_bytes, mem, nmem, scratch, status);
    hipLaunchKernelGGL(k_inflate_place<false>, dim3(nmem), dim3(IW), 0, st, mem, nmem, text, scratch, status, 1u, (size_t)0);
    hipLaunchKernelGGL(k_crc32_members, dim3((nmem + 3) / 4), dim3(256), 0, st, mem, nmem, text, comp, status);
    return hipGetLastError();
}
// the stream kernels (rk_gunzip.hip)
static hipError_t stream_lanes_attr() {
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
    if (!((attr_set.load(std::memory_order_acquire) >> dev) & 1ull) || dev == 63) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_inflate_lanes<8, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LaneLds<8, 6>));
        if (attr != hipSuccess) return attr;
        attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    return hipSuccess;
}
hipError_t launch_gz_find_starts(const uint8_t* comp, uint32_t nbits, const uint32_t* from, const uint32_t* to, uint32_t n, uint32_t* found, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gz_find_starts, dim3(n), dim3(256), 0, st, comp, nbits, from, to, found);
    return hipGetLastError();
}
hipError_t launch_gz_lanes(const uint8_t* comp, uint32_t comp_bytes, GzChunk* chunks, uint32_t n, uint32_t* scratch, hipStream_t st) {
    if (!n) return hipSuccess;
    const hipError_t a = stream_lanes_attr();
    if (a != hipSuccess) return a;
    hipLaunchKernelGGL((k_inflate_lanes<8, 6, true>), dim3((n + IW - 1) / IW), dim3(IW), sizeof(LaneLds<8, 6>), st, comp, comp_bytes, chunks, n, scratch, nullptr);
    return hipGetLastError();
}
hipError_t launch_gz_place(const GzChunk* chunks, uint32_t n, uint32_t group, const GzChunk* units, uint32_t nunits, uint8_t* planes, size_t plane_stride, const uint32_t* scratch,
                           uint8_t* rings, uint32_t* heads, uint8_t* text, uint32_t text_bytes, uint32_t* crc, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_inflate_place<true>, dim3(nunits, 3), dim3(IW), 0, st, chunks, n, planes, scratch, nullptr, group, plane_stride);
    hipLaunchKernelGGL(k_gz_windows, dim3(1), dim3(1024), 0, st, units, nunits, planes, plane_stride, rings, heads);
    hipLaunchKernelGGL(k_gz_resolve, dim3(nunits, 4), dim3(256), 0, st, units, planes, plane_stride, rings, heads, text);
    if (text_bytes) hipLaunchKernelGGL(k_crc32_segments, dim3((((text_bytes + 65535u) >> 16) + 3u) / 4u), dim3(256), 0, st, text, text_bytes, crc);
    return hipGetLastError();
}
// (code objects are loaded at a TU's first launch -- tens of milliseconds; rk_warm_up asks for a kernel's attributes ahead of time instead)
hipError_t warm_inflate() {
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_inflate_lanes<8, 6, false>));
    if (e == hipSuccess) e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_inflate_place<false>));
    if (e == hipSuccess) e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_crc32_members));
    return e;
}
hipError_t launch_fastq_first_start(const uint8_t* text, uint32_t n, uint32_t from, uint32_t window, bool at_eof, uint32_t* cuts, int which, hipStream_t st) {
    hipLaunchKernelGGL(k_fastq_first_start, dim3(1), dim3(256), 0, st, text, n, from, window, at_eof ? 1u : 0u, cuts, which);
    return hipGetLastError();
}

} // namespace rk
