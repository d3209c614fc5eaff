// rk_device.hpp -- gfx950 device building blocks shared by the rkmh_amd kernels.
//
// Replaces (on the device) what mkmh::calc_hashes does per k-mer for the call sites
// /root/reference/src/rkmh.cpp:821,860: canonical MurmurHash3_x64_128 (seed 42) of a k-mer window,
// 0 for any window holding a non-ACGT base.  Windows are read from an LDS copy of the upper-cased
// sequence and of its reverse complement, so a hash costs two unaligned 16-byte LDS window reads
// (ds_read_b128 at any byte address: gfx950 runs with unaligned DS access) and two murmur evaluations --
// no per-window complementing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rk {

struct DevPolicy {
    int32_t fold;
    int32_t drop_last_window;
    int32_t counter_counts_zero;
    int32_t mask_strict_less;
    int32_t freq_max_inclusive;
    uint32_t seed;
};

constexpr uint64_t MM_C1 = 0x87c37b91114253d5ULL;
constexpr uint64_t MM_C2 = 0x4cf5ad432745937fULL;

// 64-bit rotate by a compile-time amount as two v_alignbit_b32 (hipcc builds most of these rotates from 64-bit shifts and
// ORs, three to four instructions each); r = 33 is the half swap (free: register naming) followed by a rotate by 1
// The halves are joined by a vector bitcast, not by (hi << 32) | lo: hipcc turns that OR into a 64-bit ADD of {lo, 0} and
// {0, hi}, fuses it with the addition that follows the rotate, and pays a v_mov and an extra v_lshl_add_u64 for it.
typedef uint32_t rk_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint64_t join64(uint32_t lo, uint32_t hi) {
    const rk_u32x2 v = {lo, hi};
    return __builtin_bit_cast(uint64_t, v);
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) {
    const rk_u32x2 v = __builtin_bit_cast(rk_u32x2, x);
    uint32_t lo = v.x, hi = v.y;
    if (r >= 32) { const uint32_t t = lo; lo = hi; hi = t; r -= 32; }
    if (r == 0) return join64(lo, hi);
    const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - r);
    const uint32_t nl = __builtin_amdgcn_alignbit(lo, hi, 32 - r);
    return join64(nl, nh);
}
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33; return k;
}
// x * 5 as (x << 2) + x in ONE v_lshl_add_u64 (hipcc otherwise spends two v_mad_u64_u32 and a move on it)
__device__ __forceinline__ uint64_t mul5(uint64_t x) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ void mm_block(uint64_t& h1, uint64_t& h2, uint64_t k1, uint64_t k2) {
    k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = mul5(h1) + 0x52dce729;
    k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = mul5(h2) + 0x38495ab5;
}
// First block of a hash: h1 = h2 = seed are wave-uniform, so "h1 += h2" folds into the constant of the multiply-add that
// follows ((h1 + seed) * 5 + c = h1 * 5 + (5 * seed + c)): one 64-bit add less per strand.
__device__ __forceinline__ void mm_block_first(uint64_t& h1, uint64_t& h2, uint32_t seed, uint64_t k1, uint64_t k2) {
    k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2;
    h1 = rotl64(k1 ^ (uint64_t)seed, 27);
    h1 = mul5(h1) + ((uint64_t)seed * 5u + 0x52dce729u);
    k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1;
    h2 = rotl64(k2 ^ (uint64_t)seed, 31); h2 += h1; h2 = mul5(h2) + 0x38495ab5;
}
// finalisation + the 128->64 fold (policy U1).  FOLD >= 0 fixes the fold at compile time (no branches
// between the forward and reverse-complement hash chains, so the scheduler can interleave them).
template <int FOLD = -1>
__device__ __forceinline__ uint64_t mm_finish(uint64_t h1, uint64_t h2, uint32_t len, int fold_rt) {
    const int fold = FOLD >= 0 ? FOLD : fold_rt;
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    if (FOLD == 3) return h1;                        // fold 0 with the half swap left to min_swapped()
    if (fold == 0) return (h1 << 32) | (h1 >> 32);   // ((u64)w[0] << 32) | w[1]
    if (fold == 1) return h1;                        // *(u64*)w
    h2 += h1;                                        // ((u64)w[2] << 32) | w[1]
    return (h2 << 32) | (h1 >> 32);
}


// Canonical hash under fold 0 (the half-swapped h1) from the two UNSWAPPED h1 values: the smaller of swap(f), swap(r).
// Comparing and selecting the halves directly saves the two v_mov per strand that materialising swap() into an aligned
// register pair for v_cmp_lt_u64 costs.
__device__ __forceinline__ uint64_t min_swapped(uint64_t f, uint64_t r) {
    const rk_u32x2 a = __builtin_bit_cast(rk_u32x2, f), b = __builtin_bit_cast(rk_u32x2, r);
    const bool lt = a.x < b.x || (a.x == b.x && a.y < b.y); // .x (low half of h1) is the high half of the folded value
    return join64(lt ? a.y : b.y, lt ? a.x : b.x);
}

// 2-bit base code of an upper-case base: (ascii >> 1) & 3  =>  A=0, C=1, T=2, G=3; complement = code ^ 2.
// Packed k-mer: base i of the string in bits [2i, 2i+2).
// ---- forward-strand group filter of the k-mer-space kernel (rk_kmer.hip) --------------------------------------------------------
// One lane of that kernel examines FOUR consecutive windows of a read (a "group": read positions 4g .. 4g+3), which it reads from
// the packed image as one super-window of k + 3 bases.  All four windows contain the (k-3)-mer at super-window offset 3 (the
// "core"), so one 16-byte SECTOR of the filter, chosen by the core, serves the whole group: dword j of the sector holds the bit
// pairs of every found k-mer X whose alignment-j core (X >> 2 * (3 - j)) maps to the sector.  Keys are entered in BOTH orientations
// (X and its reverse complement), so the window loop never forms a reverse complement; the drain canonicalises the few candidates.
__host__ __device__ __forceinline__ uint32_t kf4_core_mask(int k) { return k > 3 ? (1u << (2 * (k - 3))) - 1u : 0u; }
// sector of a core among n sectors (any n, not only powers of two: the filter is sized against the L2 in steps finer than x 2):
// the hashed core scaled to [0, n) by the high half of a 32 x 32 product (for n = 2^lg this is the old `>> (32 - lg)`)
__host__ __device__ __forceinline__ uint32_t kf4_sector(uint32_t core, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(core * 0x85EBCA6Bu, n);
#else
    return (uint32_t)(((uint64_t)(core * 0x85EBCA6Bu) * (uint64_t)n) >> 32);
#endif
}
// The bits of a k-mer inside its dword.  RK_KF4_MID = 1 (round 4): three five-bit numbers out of the MIDDLE of the 64-bit product
// p = x * odd constant -- bits 24..28 (byte 3 of the low half), 32..36 and 48..52 (bytes 0 and 2 of the high half).  Byte-aligned
// fields cost one instruction each on the device (v_lshlrev_b32 takes the low five bits of its shift operand, SDWA selects the
// byte: 1 << field without a shift or a mask in front), the whole test is v_mad_u64_u32 + 3 shifts + v_or3 + v_and + v_cmp = 7
// instructions against 9 for three v_bfe of the top fifteen bits of the low half (RK_KF4_MID = 0, rounds 2-3) -- and the middle
// bits mix better: 3.9 false candidates per C2 read against 4.1 (tools/kf4_fp_model.py, which also shows what does NOT work:
// bytes 3,2,1 of the high half alone -- nearly linear in the top bases of x -- pass 6.1).
#ifndef RK_KF4_MID
#define RK_KF4_MID 1
#endif
#ifndef RK_KF4_NBITS
#define RK_KF4_NBITS 3 // bits per entry.  3 against 2 (measured, each at its best density): C2 0.314 / 0.321 ms, 266 references 0.332 / 0.343,
                       // s = 2000 0.626 / 0.646, 400 references 0.407 / 0.397: fewer false candidates per byte of filter for one more bit test per window
#endif
__host__ __device__ __forceinline__ uint32_t kf4_h(uint32_t x) { return x * 0x9E3779B1u; }
__host__ __device__ __forceinline__ uint32_t kf4_bits(uint32_t x) {
#if RK_KF4_MID
    const uint64_t p = (uint64_t)x * 0x9E3779B1ull;
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32);
    uint32_t b = (1u << ((lo >> 24) & 31u)) | (1u << (hi & 31u));
    if (RK_KF4_NBITS >= 3) b |= 1u << ((hi >> 16) & 31u);
    return b;
#else
    const uint32_t h = kf4_h(x);
    uint32_t b = (1u << (h >> 27)) | (1u << ((h >> 22) & 31u));
    if (RK_KF4_NBITS >= 3) b |= 1u << ((h >> 17) & 31u);
    return b;
#endif
}
// kf4_bits on the device (same value; host compilation pass: the portable form)
__device__ __forceinline__ uint32_t kf4_bits_dev(uint32_t x) {
#if RK_KF4_MID && defined(__HIP_DEVICE_COMPILE__)
    const uint64_t p = (uint64_t)x * 0x9E3779B1ull;
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32), one = 1u;
    uint32_t b3, b2 = 0u;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(b3) : "v"(lo), "v"(one));
    if (RK_KF4_NBITS >= 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(b2) : "v"(hi), "v"(one));
    return b3 | (1u << (hi & 31u)) | b2;
#else
    return kf4_bits(x);
#endif
}

// reverse complement of a packed k-mer (k <= 16)
__host__ __device__ __forceinline__ uint32_t packed_revcomp(uint32_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r = __builtin_bitreverse32(v);
#else
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
#endif
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);  // 2-bit groups in reverse order, bits of a group in order
    if (k < 16) r >>= 2 * (16 - k);
    const uint32_t m = k < 16 ? ((1u << (2 * k)) - 1u) : 0xffffffffu;
    return (r ^ 0xAAAAAAAAu) & m;
}

// ---- wide k-mers of the k-mer-space kernel: k = 17 .. KW_MAX_K (34 .. 40 bits, held in 64) ------------------------------------
constexpr int KW_MAX_K = 20;
__host__ __device__ __forceinline__ uint64_t packed_revcomp64(uint64_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r = __builtin_bitreverse64(v);
#else
    uint64_t r = 0;
    for (int i = 0; i < 64; ++i) r |= ((v >> i) & 1ull) << (63 - i);
#endif
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1); // 2-bit groups in reverse order, bits of a group in order
    if (k < 32) r >>= 2 * (32 - k);
    const uint64_t m = k < 32 ? ((1ull << (2 * k)) - 1ull) : ~0ull;
    return (r ^ 0xAAAAAAAAAAAAAAAAull) & m;
}
// 32 bits out of a wide k-mer (or core) for the filter's sector and bit choices: the same function on host and device
__host__ __device__ __forceinline__ uint32_t kw_fold(uint64_t x) { return (uint32_t)x ^ ((uint32_t)(x >> 32) * 0x85EBCA6Bu + 0x7F4A7C15u); }
// The exact map of wide k-mers (km2): y = (key * odd constant) mod 4^k is a bijection on the 2k-bit k-mers; bucket = its top km1_b bits,
// tag = the next KW_TAG bits.  A bucket is four 4-byte cells hop:3 | tag:8 | flag:1 | key number:20 (flag in the last cell as in
// km1; key number all ones = empty); the key number leads to kkeys[number] = {k-mer bits 0..31, k-mer bits 32..39 | value id << 8}
// (8 bytes: the table shares the L2 with the map and the filter): the full k-mer settles what the 8-bit tag could not, the value
// id is km1's (below the reference count: that reference once; KW_VID_ZERO: canonical hash 0; else nref + compound index);
// kslots[number] = the index key id, read only under the per-key mask of -M.
constexpr uint64_t KW_C = 0x9E3779B97F4A7C15ull;
constexpr uint32_t KW_TAG = 8, KW_IDBITS = 20, KW_EMPTY = (1u << KW_IDBITS) - 1u, KW_VID_ZERO = 0xFFFFFEu, KW_VID_NONE = 0xFFFFFFu; // (value ids are 24 bits)
__host__ __device__ __forceinline__ uint64_t kw_y(uint64_t key, int k) { return (key * KW_C) & ((1ull << (2 * k)) - 1ull); }

// MurmurHash3_x64_128 of a k-mer of k <= 16 bytes held in four dwords (little-endian byte order, bytes beyond k zero)
template <int FOLD = -1>
__device__ __forceinline__ uint64_t murmur_regs16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int k, uint32_t seed, int fold) {
    uint64_t h1 = seed, h2 = seed;
    if (k == 16) mm_block_first(h1, h2, seed, join64(w0, w1), join64(w2, w3));
    else { // tail only, as in murmur_window
        uint64_t k1 = join64(w0, w1), k2 = join64(w2, w3);
        if (k > 8) { k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2; }
        k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    }
    return mm_finish<FOLD>(h1, h2, (uint32_t)k, fold);
}
// the four ASCII bases 'A','C','T','G' of the 2-bit codes in byte `d` of a packed k-mer
__device__ __forceinline__ uint32_t packed_to_ascii4(uint32_t v, int d) {
    const uint32_t b = (v >> (8 * d)) & 0xffu;
    const uint32_t sel = (b & 3u) | ((b & 0xCu) << 6) | ((b & 0x30u) << 12) | ((b & 0xC0u) << 18);
    return __builtin_amdgcn_perm(0x47544341u, 0x47544341u, sel); // code 0..3 -> "ACTG"
}
// canonical hash (min over both strands of the folded value) of a packed k-mer, k <= 16
__device__ __forceinline__ uint64_t canonical_packed(uint32_t v, int k, uint32_t seed, int fold) {
    const uint32_t rv = packed_revcomp(v, k);
    uint32_t f[4], r[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int nv = k - 4 * d; // bytes of this dword that belong to the k-mer
        const uint32_t m = nv >= 4 ? 0xffffffffu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
        f[d] = packed_to_ascii4(v, d) & m;
        r[d] = packed_to_ascii4(rv, d) & m;
    }
    const uint64_t hf = murmur_regs16<-1>(f[0], f[1], f[2], f[3], k, seed, fold);
    const uint64_t hr = murmur_regs16<-1>(r[0], r[1], r[2], r[3], k, seed, fold);
    return hf < hr ? hf : hr;
}

// canonical hash of a wide packed k-mer (16 < k <= 20): one 16-byte block and a tail of k - 16 bytes per strand
__device__ __forceinline__ uint64_t murmur_regs20(const uint32_t* w, int k, uint32_t seed, int fold) {
    uint64_t h1 = seed, h2 = seed;
    mm_block_first(h1, h2, seed, join64(w[0], w[1]), join64(w[2], w[3]));
    uint64_t k1 = (uint64_t)w[4]; // (k - 16 <= 4 tail bytes: k2 stays 0)
    k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    return mm_finish<-1>(h1, h2, (uint32_t)k, fold);
}
__device__ __forceinline__ uint32_t packed_to_ascii4_64(uint64_t v, int d) {
    const uint32_t b = (uint32_t)(v >> (8 * d)) & 0xffu;
    const uint32_t sel = (b & 3u) | ((b & 0xCu) << 6) | ((b & 0x30u) << 12) | ((b & 0xC0u) << 18);
    return __builtin_amdgcn_perm(0x47544341u, 0x47544341u, sel);
}
__device__ __forceinline__ uint64_t canonical_packed64(uint64_t v, int k, uint32_t seed, int fold) {
    const uint64_t rv = packed_revcomp64(v, k);
    uint32_t f[5], r[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const int nv = k - 4 * d;
        const uint32_t m = nv >= 4 ? 0xffffffffu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
        f[d] = packed_to_ascii4_64(v, d) & m;
        r[d] = packed_to_ascii4_64(rv, d) & m;
    }
    const uint64_t hf = murmur_regs20(f, k, seed, fold);
    const uint64_t hr = murmur_regs20(r, k, seed, fold);
    return hf < hr ? hf : hr;
}

// MurmurHash3_x64_128 of the k bytes starting at byte offset `a` of the LDS dword array w32.
// The array must be readable for 16 bytes past the window (buffers are padded).
// Windows are fetched with UNALIGNED 16-byte LDS reads: gfx950 runs with unaligned DS access enabled (hipcc itself
// emits one ds_read_b128 for an align-1 vector load; verified on MI355X by tools/ubench/ubench_unaligned_lds.hip), so no
// dword-aligned reads + v_alignbyte funnel is needed.
typedef uint32_t rk_u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) rk_unaligned16 { rk_u32x4 v; };
__device__ __forceinline__ rk_u32x4 lds_load16_unaligned(const uint32_t* w32, uint32_t byte_off) {
    return reinterpret_cast<const rk_unaligned16*>(reinterpret_cast<const uint8_t*>(w32) + byte_off)->v;
}
template <int KT, int FOLD = -1>
__device__ __forceinline__ uint64_t murmur_window(const uint32_t* w32, uint32_t a, int k_rt, uint32_t seed, int fold) {
    const int k = KT ? KT : k_rt;
    uint64_t h1 = seed, h2 = seed;
    const int nblocks = k >> 4;
    for (int b = 0; b < nblocks; ++b) {
        const rk_u32x4 w = lds_load16_unaligned(w32, a);
        if (b == 0) mm_block_first(h1, h2, seed, join64(w.x, w.y), join64(w.z, w.w));
        else mm_block(h1, h2, join64(w.x, w.y), join64(w.z, w.w));
        a += 16;
    }
    const int rem = k & 15;
    if (rem) {
        const rk_u32x4 w = lds_load16_unaligned(w32, a);
        uint32_t t[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int nv = rem - 4 * q;
            uint32_t m = nv >= 4 ? 0xffffffffu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
            t[q] &= m;
        }
        uint64_t k1 = (uint64_t)t[0] | ((uint64_t)t[1] << 32);
        uint64_t k2 = (uint64_t)t[2] | ((uint64_t)t[3] << 32);
        if (rem > 8) { k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2; }
        k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    }
    return mm_finish<FOLD>(h1, h2, (uint32_t)k, fold);
}

// Canonical hash of a window whose length is only known at run time: both strands advance through ONE loop over the
// 16-byte blocks (the two chains interleave, half the loop control of two separate calls) and the tail is masked with
// wave-uniform masks prepared once per kernel (TailMasks) instead of per-window shift arithmetic.
struct TailMasks { uint32_t m[4]; uint32_t rem; };
__device__ __forceinline__ TailMasks make_tail_masks(int k) {
    TailMasks t;
    t.rem = (uint32_t)k & 15u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nv = (int)t.rem - 4 * q;
        t.m[q] = nv >= 4 ? 0xffffffffu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
    }
    return t;
}
// NB >= 0: the number of whole 16-byte blocks is known at compile time (the block loop unrolls and the first block takes the
// constant-folded form); NB < 0: any number of blocks
template <int NB>
__device__ __forceinline__ uint64_t canonical_nb(const uint32_t* fwd, uint32_t af, const uint32_t* rc, uint32_t ar, int k,
                                                 const TailMasks& tm, uint32_t seed, int fold) {
    uint64_t f1 = seed, f2 = seed, r1 = seed, r2 = seed;
    if constexpr (NB >= 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const rk_u32x4 wf = lds_load16_unaligned(fwd, af);
            const rk_u32x4 wr = lds_load16_unaligned(rc, ar);
            if (b == 0) {
                mm_block_first(f1, f2, seed, join64(wf.x, wf.y), join64(wf.z, wf.w));
                mm_block_first(r1, r2, seed, join64(wr.x, wr.y), join64(wr.z, wr.w));
            } else {
                mm_block(f1, f2, join64(wf.x, wf.y), join64(wf.z, wf.w));
                mm_block(r1, r2, join64(wr.x, wr.y), join64(wr.z, wr.w));
            }
            af += 16; ar += 16;
        }
    } else {
        for (int b = 0; b < (k >> 4); ++b) {
            const rk_u32x4 wf = lds_load16_unaligned(fwd, af);
            const rk_u32x4 wr = lds_load16_unaligned(rc, ar);
            mm_block(f1, f2, join64(wf.x, wf.y), join64(wf.z, wf.w));
            mm_block(r1, r2, join64(wr.x, wr.y), join64(wr.z, wr.w));
            af += 16; ar += 16;
        }
    }
    if (tm.rem) {
        const rk_u32x4 wf = lds_load16_unaligned(fwd, af);
        const rk_u32x4 wr = lds_load16_unaligned(rc, ar);
        uint64_t fk1 = join64(wf.x & tm.m[0], wf.y & tm.m[1]);
        uint64_t rk1 = join64(wr.x & tm.m[0], wr.y & tm.m[1]);
        if (tm.rem > 8) {
            uint64_t fk2 = join64(wf.z & tm.m[2], wf.w & tm.m[3]);
            uint64_t rk2 = join64(wr.z & tm.m[2], wr.w & tm.m[3]);
            fk2 *= MM_C2; fk2 = rotl64(fk2, 33); fk2 *= MM_C1; f2 ^= fk2;
            rk2 *= MM_C2; rk2 = rotl64(rk2, 33); rk2 *= MM_C1; r2 ^= rk2;
        }
        fk1 *= MM_C1; fk1 = rotl64(fk1, 31); fk1 *= MM_C2; f1 ^= fk1;
        rk1 *= MM_C1; rk1 = rotl64(rk1, 31); rk1 *= MM_C2; r1 ^= rk1;
    }
    const uint64_t f = mm_finish<-1>(f1, f2, (uint32_t)k, fold);
    const uint64_t r = mm_finish<-1>(r1, r2, (uint32_t)k, fold);
    return f < r ? f : r;
}
__device__ __forceinline__ uint64_t canonical_rt(const uint32_t* fwd, uint32_t af, const uint32_t* rc, uint32_t ar, int k,
                                                 const TailMasks& tm, uint32_t seed, int fold) {
    // k is wave-uniform: one scalar branch picks the unrolled form of the common sizes (k = 16 .. 31: one block, 32 .. 47: two)
    const int nb = k >> 4;
    if (nb == 1) return canonical_nb<1>(fwd, af, rc, ar, k, tm, seed, fold);
    if (nb == 2) return canonical_nb<2>(fwd, af, rc, ar, k, tm, seed, fold);
    if (nb == 0) return canonical_nb<0>(fwd, af, rc, ar, k, tm, seed, fold);
    return canonical_nb<-1>(fwd, af, rc, ar, k, tm, seed, fold);
}

// ---- per-dword (4 bases) SWAR helpers -------------------------------------------------------
// mkmh::to_upper quirk: every (signed) char > 91 gets -32 (bytes >= 128 are negative => untouched)
__device__ __forceinline__ uint32_t upper4(uint32_t x) {
    uint32_t t = (x & 0x7f7f7f7fu) + 0x24242424u;   // bit7 <=> low7 >= 92
    uint32_t m = t & ~x & 0x80808080u;
    return x - (m >> 2);
}
// Letter tables indexed by bits 2:1 of an upper-case base (A=0, C=1, T=2, G=3): one v_perm_b32 looks four bases up.
// acgt_mismatch4: byte q is non-zero <=> byte q of x is NOT one of 'A','C','G','T'.
__device__ __forceinline__ uint32_t acgt_mismatch4(uint32_t x) {
    const uint32_t sel = (x >> 1) & 0x03030303u;
    return x ^ __builtin_amdgcn_perm(0x47544341u, 0x47544341u, sel); // "ACTG"
}
// reverse complement of four upper-case bases (bytes that are not A/C/G/T: don't care, their windows are never hashed)
__device__ __forceinline__ uint32_t revcomp4(uint32_t x) {
    const uint32_t sel = (x >> 1) & 0x03030303u;
    const uint32_t c = __builtin_amdgcn_perm(0x43414754u, 0x43414754u, sel); // "TGAC": A->T, C->G, T->A, G->C
    return __builtin_bswap32(c);
}
// one dword at any byte offset of an LDS dword array (unaligned DS access, see murmur_window below)
struct __attribute__((packed)) rk_unaligned4 { uint32_t v; };
__device__ __forceinline__ uint32_t lds_load4_unaligned(const uint32_t* w32, uint32_t byte_off) {
    return reinterpret_cast<const rk_unaligned4*>(reinterpret_cast<const uint8_t*>(w32) + byte_off)->v;
}
// 4-bit mask: bit q set <=> byte q is NOT one of 'A','C','G','T'
__device__ __forceinline__ uint32_t invalid4(uint32_t x) {
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t b = (x >> (8 * q)) & 0xffu;
        uint32_t ok = ((b >> 5) == 2u) ? ((0x0010008Au >> (b & 31u)) & 1u) : 0u;
        r |= (ok ^ 1u) << q;
    }
    return r;
}

// LDS image of one staged sequence piece.
//   fwd  : dword array; the piece's first base sits at byte offset FWD_PAD + d (d = global misalignment)
//   rc   : dword array; reverse complement string, byte 0 = complement of the LAST base
//   inv  : bit (FWD_PAD + d + t) set <=> base t invalid (bits indexed by fwd byte position)
constexpr int FWD_PAD = 4;   // bytes in front of the fwd image so that p-3 reads stay in bounds
constexpr int TAIL_PAD = 40; // bytes readable past either string (murmur_window over-read)

__host__ __device__ constexpr int stage_lds_dwords(int max_bases) {
    // fwd + rc + inv
    return ((FWD_PAD + 3 + max_bases + TAIL_PAD + 3) / 4) + ((max_bases + TAIL_PAD + 3) / 4) +
           ((FWD_PAD + 3 + max_bases + 31) / 32 + 2);
}

struct Staged {
    uint32_t* fwd;
    uint32_t* rc;
    uint32_t* inv;
    uint32_t fbase; // byte offset of base 0 inside fwd
    uint32_t nbases;
};

// Cooperative staging by a group of G threads (tid in [0,G)), G a multiple of 8 and all threads of
// the group call it.  `sync` is __syncthreads (block groups) -- for one-wave groups the block IS the wave.
template <typename SyncFn>
__device__ __forceinline__ Staged stage_piece(const uint8_t* __restrict__ bases, uint64_t start, uint32_t nbases,
                                              uint32_t* lds, int max_bases, int tid, int G, SyncFn sync) {
    Staged s;
    const int fwd_dw = (FWD_PAD + 3 + max_bases + TAIL_PAD + 3) / 4;
    const int rc_dw = (max_bases + TAIL_PAD + 3) / 4;
    s.fwd = lds;
    s.rc = lds + fwd_dw;
    s.inv = s.rc + rc_dw;
    const uint32_t d = (uint32_t)(start & 3);
    s.fbase = FWD_PAD + d;
    s.nbases = nbases;
    const uint32_t* g32 = reinterpret_cast<const uint32_t*>(bases) + (start >> 2);
    const uint32_t ndw = (d + nbases + 3) >> 2;       // global dwords covering the piece
    // fwd dword j (j>=1) holds global dword j-1 (FWD_PAD = 4 bytes = 1 dword)
    for (uint32_t j0 = 0; j0 < ndw + 1; j0 += G) {
        uint32_t j = j0 + tid;                         // fwd dword index
        uint32_t x = 0;
        if (j >= 1 && j <= ndw) x = upper4(g32[j - 1]);
        uint32_t nib = invalid4(x);
        if (j <= ndw) s.fwd[j] = x;
        uint32_t n = nib << (4 * (tid & 7));
        n |= __shfl_xor((int)n, 1);
        n |= __shfl_xor((int)n, 2);
        n |= __shfl_xor((int)n, 4);
        if ((tid & 7) == 0 && (j >> 3) <= (ndw >> 3)) s.inv[j >> 3] = n;  // never past the word holding dword ndw
    }
    sync();
    // rc dword q covers rc bytes 4q..4q+3 = complement of fwd bytes p+3..p, p = fbase + nbases - 4 - 4q
    const uint32_t nrc = (nbases + 3) >> 2;
    for (uint32_t q = tid; q < nrc; q += G) {
        const uint32_t p = s.fbase + nbases - 4u - 4u * q; // >= FWD_PAD + d - 3 >= 1
        s.rc[q] = revcomp4(lds_load4_unaligned(s.fwd, p));
    }
    sync();
    return s;
}

// all k bases of window i valid?  (bits fbase+i .. fbase+i+k-1 of inv all zero)
template <int KT>
__device__ __forceinline__ bool window_valid(const Staged& s, uint32_t i, int k_rt) {
    const int k = KT ? KT : k_rt;
    uint32_t bit = s.fbase + i;
    uint32_t idx = bit >> 5, sh = bit & 31;
    int left = k;
    uint32_t lo = s.inv[idx];
    uint32_t acc = 0;
    while (left > 0) {
        uint32_t hi = s.inv[idx + 1];
        uint32_t x = __builtin_amdgcn_alignbit(hi, lo, sh); // 32 bits starting at `bit`
        uint32_t m = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        acc |= x & m;
        lo = hi; ++idx; left -= 32;
    }
    return acc == 0;
}

// canonical hash of window i (k-mer size k) of a staged piece
template <int KT>
__device__ __forceinline__ uint64_t canonical_window(const Staged& s, uint32_t i, int k_rt, const DevPolicy& pol) {
    const int k = KT ? KT : k_rt;
    if (!window_valid<KT>(s, i, k)) return 0;
    uint64_t f = murmur_window<KT>(s.fwd, s.fbase + i, k, pol.seed, pol.fold);
    uint64_t r = murmur_window<KT>(s.rc, s.nbases - (uint32_t)k - i, k, pol.seed, pol.fold);
    return f < r ? f : r;
}

// ---- canonical murmur3 of a k-mer given by a byte functor (k-mers read from global memory or built on the fly) ----
template <typename GetByte>
__device__ __forceinline__ uint64_t murmur_bytes(GetByte gb, int k, uint32_t seed, int fold) {
    uint64_t h1 = seed, h2 = seed;
    const int nblocks = k >> 4;
    int p = 0;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1 = 0, k2 = 0;
        for (int q = 0; q < 8; ++q) k1 |= (uint64_t)gb(p + q) << (8 * q);
        for (int q = 0; q < 8; ++q) k2 |= (uint64_t)gb(p + 8 + q) << (8 * q);
        mm_block(h1, h2, k1, k2);
        p += 16;
    }
    const int rem = k & 15;
    if (rem) {
        uint64_t k1 = 0, k2 = 0;
        for (int q = 0; q < rem && q < 8; ++q) k1 |= (uint64_t)gb(p + q) << (8 * q);
        for (int q = 8; q < rem; ++q) k2 |= (uint64_t)gb(p + q) << (8 * (q - 8));
        if (rem > 8) { k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2; }
        k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    }
    return mm_finish<-1>(h1, h2, (uint32_t)k, fold);
}
__device__ __forceinline__ bool is_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
__device__ __forceinline__ uint8_t comp1(uint8_t c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }
template <typename GetByte>
__device__ __forceinline__ uint64_t canonical_bytes(GetByte gb, int k, const DevPolicy& pol) {
    for (int q = 0; q < k; ++q) if (!is_acgt(gb(q))) return 0;
    const uint64_t f = murmur_bytes(gb, k, pol.seed, pol.fold);
    const uint64_t r = murmur_bytes([&](int q) -> uint8_t { return comp1(gb(k - 1 - q)); }, k, pol.seed, pol.fold);
    return f < r ? f : r;
}
// mkmh::to_upper on one byte: every (signed) char > 91 gets -32 (see upper4)
__device__ __forceinline__ uint8_t upper1(uint8_t c) { return (c > 91 && c < 128) ? (uint8_t)(c - 32) : c; }

__host__ __device__ __forceinline__ int num_windows(int len, int k, int drop_last) {
    int n = drop_last ? len - k : len - k + 1;
    return n > 0 ? n : 0;
}

// h % slots for the -M table (HASHTCounter slot, rkmh.cpp:739) without a 64-bit division: m = floor((2^64 - 1) / slots) comes
// from the host; q = mulhi(h, m) is the quotient or one less (h * m / 2^64 lies in (h / slots - 1, h / slots]), so one conditional
// subtraction finishes it.  Exact for every h and every slots >= 1 (2 * slots < 2^64).  ~20 VALU instead of ~55.
__device__ __forceinline__ uint64_t mod_slots(uint64_t h, uint64_t slots, uint64_t m) {
    const uint64_t q = __umul64hi(h, m);
    uint64_t r = h - q * slots;
    if (r >= slots) r -= slots;
    return r;
}

// ---- resident reference index ------------------------------------------------------------------
// All distinct hashes of all reference sketches in one bucketed hash table sized to stay L2 resident:
//   fpb  : one 16-byte bucket = 8 x 16-bit fingerprints (0 = empty; slots of a bucket fill in order)
//   base : per bucket, keys stored before it; key id = base[bucket] + position in the bucket
//   kv   : per key id {full 64-bit hash (read only to confirm a fingerprint match), value, pad} in one 16-byte entry
//   value: per key, bit31=1 -> offset into post; else bits 30:29 = 0: one posting inline (ref | mult<<20, mult<512),
//          1: two postings of multiplicity 1 inline (ref1 | ref2<<11, both < 2048)
//   post : [off] = count, then count x (ref, mult)
// A lookup that misses (about 7 of 8 read k-mers) costs exactly one 16-byte load.  Eight narrow fingerprints per
// bucket instead of four wide ones halve the table for the same number of keys (1 MB at C2, 2 MB for ~300 references
// against the 4 MB L2 of an XCD -- measured: a 4 MB table costs 20-25 % of the kernel, an 8 MB one doubles its time)
// and make bucket overflow (> 8 keys where the mean is <= 2.5) a 0.1 % event.
#ifndef RK_KMER_INLINE_N
#define RK_KMER_INLINE_N 1 // k-mer-space value table: lists of 3..6 references that hold the hash once each are stored inline (see build_index)
#endif
struct RefIndex {
    const uint4* fpb;
    const uint32_t* base; // per bucket: number of keys stored in the buckets before it => key id = base[bucket] + slot in bucket
    const uint4* kv;      // per key id {key lo, key hi, value, 0}, DENSE (no holes): a 16-byte fetch verifies a hit and brings its
                          // postings, and the whole array (16 B x distinct sketch hashes) is small enough to live in L2
    const uint32_t* post;
    uint32_t bmask;   // buckets - 1
    uint32_t bshift;  // 32 - log2(buckets)
    int32_t nref;
    // Optional first-level filter for large panels (nullptr = none): one bit pair per key in a <= 2 MB bit array that stays
    // in an XCD's L2 when the bucket table (16 B per 2.5 keys) no longer does.  The hashing loop then tests the filter word
    // and only windows that pass go to the queue; the drain looks them up in the table as it always did.
    const uint32_t* pre;
    uint32_t pmask;   // filter words - 1
    // Optional k-mer-space structures (nullptr = none; a single k from 8 to 16 only), built from the EXHAUSTIVE enumeration of the
    // 4^k k-mer universe (k_enum_kmers): every k-mer whose canonical hash is a key of this index -- or is 0 -- is in them, so they
    // have no false negatives BY CONSTRUCTION: a window that fails the filter provably hashes to a non-zero value that is in no
    // sketch, and k_classify_kmer does not hash it at all.  They are only built when every index key has exactly one preimage
    // (the enumeration checks); otherwise the hash-space kernels serve the panel.
    uint32_t kpk;     // the k they were enumerated for
    // forward-strand group filter of k_classify_kmer (rk_kmer.hip; see kf4_sector above): kf4_n sectors of 16 bytes
    const uint4* kf4;
    uint32_t kf4_n;
    const uint4* km1;      // single-probe exact map (see KM1_C above), 2^km1_b buckets
    uint32_t km1_b;
    const uint32_t* km1_vals;
    // posting lists of the k-mer-space kernel (rk_api.hip, build_kpost): identical lists stored once, lists close to one of up to
    // eight BASE lists stored as (base, exceptions); kbase = [8 x (start, members)] then the members
    const uint32_t* kpost;
    const uint32_t* kbase;
    uint32_t kbase_n;     // base lists in use (0: no list of this index is stored as (base, exceptions))
    const uint2* kkeys;   // wide k-mers (k = 17 .. 20, kpk says which): km1 then holds km2 buckets, see KW_C above
    const uint32_t* kslots;
    // -M with a bounded min_num (rk_set_min_num_bound): bit (key id) set <=> the key's slot of the depth map passes the threshold
    // (nullptr = no per-key mask).  The hash-space kernels test it for the windows that HIT a key instead of one bit of the
    // 25 MB slot bitmap for every window; the k-mer-space kernel reads a copy of km1 in which the dropped keys carry the zero id.
    const uint32_t* keepkey;
};
// The exact map the k-mer-space kernel (rk_kmer.hip) resolves its candidates in: every k-mer the enumeration found, one 16-byte
// bucket of four 4-byte cells per lookup.  y = (key * odd constant) mod 4^k is a bijection on the 2k-bit k-mers, so (bucket = top
// km1_b bits of y, remainder = the other r = 2k - km1_b bits) IS the key.  A cell, from the top: the tag (KM1_HB + r bits: how many
// buckets past its own the key was stored -- its own was full -- then the remainder), one flag bit (in the bucket's last cell: a
// key that wanted this bucket, or passed through it, lies further on) and the value id (vb = 31 - KM1_HB - r bits).  Value ids
// below the number of references are a single posting of multiplicity 1 (the id is the reference); all ones - 1 marks a k-mer
// whose canonical hash is 0; other ids index km1_vals (offset by the number of references), which holds index values in the
// RefIndex::kv format; all ones = empty.  At the load the table is built for (~0.6: 1 MB for the 161 k k-mers of the 182-genome
// panel, so that it shares an XCD's 4 MB L2 with the 2 MB filter) nine lookups in ten end in the first bucket.
constexpr uint32_t KM1_C = 0x9E3779B1u;
constexpr uint32_t KM1_HB = 3;                       // hop bits: a key lies at most 7 buckets past its own
__host__ __device__ __forceinline__ uint32_t km1_y(uint32_t key, int k) { return k >= 16 ? key * KM1_C : (key * KM1_C) & ((1u << (2 * k)) - 1u); }
__host__ __device__ __forceinline__ uint32_t km1_vbits(int k, uint32_t b) { return 31u - KM1_HB - ((uint32_t)(2 * k) - b); }
// the k-mer-space structures of EVERY k of a run with several k-mer sizes (each from 8 to 16): the multi-k form of k_classify_kmer
// walks a tile once per k (hashes of all k count towards the same per-read sketch comparison, U5) with that k's filter and map
constexpr int KM_MAX_KS = 8;
struct KmerSets {
    int32_t n;
    int32_t k[KM_MAX_KS];
    const uint4* kf4[KM_MAX_KS];
    const uint4* km1[KM_MAX_KS];
    const uint32_t* km1_vals[KM_MAX_KS];
    uint32_t kf4_n[KM_MAX_KS], km1_b[KM_MAX_KS];
};
// filter word and bit pair of a hash: the word from the low bits of the high hash word (like the bucket), the two bits
// from bits 14..23 of the low word (bits 0..13 are the fingerprint)
__host__ __device__ __forceinline__ uint32_t index_pre_word(uint64_t h, uint32_t pmask) { return (uint32_t)(h >> 32) & pmask; }
__host__ __device__ __forceinline__ uint32_t index_pre_bits(uint64_t h) {
    const uint32_t lo = (uint32_t)h;
    return (1u << ((lo >> 14) & 31u)) | (1u << ((lo >> 19) & 31u));
}
constexpr uint32_t IDX_NOT_FOUND = 0xffffffffu;
constexpr int IDX_SLOTS = 8;

// fingerprint = 14 hash bits + the "occupied" bit 14 (never 0).  Bit 15 of a bucket's slot-0 halfword is the bucket's
// OVERFLOW flag: some key that hashed here (or passed through here) was stored further down the chain.  Only then
// does a lookup that found no fingerprint match have to look at the next bucket.  (A false fingerprint match -- 2.5
// keys x 2^-14 per lookup -- only costs a verification in the drain: exactness comes from the 64-bit key compare.)
constexpr uint32_t IDX_OVF = 0x8000u;
__host__ __device__ __forceinline__ uint32_t index_fp(uint64_t h) { return ((uint32_t)h & 0x3fffu) | 0x4000u; }
// bucket = the LOW bits of the hash's high word (the fingerprint takes the low word's bits).  Murmur output is already
// uniform, so no further mixing is spent in the hot loop -- but the keys are bottom-S sketch hashes, i.e. the SMALLEST
// hashes of their sequences, so the top bits of the 64-bit value are heavily biased and must not be used.
__host__ __device__ __forceinline__ uint32_t index_bucket(uint64_t h, uint32_t bmask) {
    return (uint32_t)(h >> 32) & bmask;
}
// 8-bit mask of the bucket's slots whose fingerprint equals fp (slot 0's overflow flag ignored)
__device__ __forceinline__ uint32_t index_match_mask(const uint4& f, uint32_t fp) {
    return (((f.x & 0x7fffu) == fp) ? 1u : 0u) | (((f.x >> 16) == fp) ? 2u : 0u) |
           (((f.y & 0xffffu) == fp) ? 4u : 0u) | (((f.y >> 16) == fp) ? 8u : 0u) |
           (((f.z & 0xffffu) == fp) ? 16u : 0u) | (((f.z >> 16) == fp) ? 32u : 0u) |
           (((f.w & 0xffffu) == fp) ? 64u : 0u) | (((f.w >> 16) == fp) ? 128u : 0u);
}
__device__ __forceinline__ uint64_t index_key(const RefIndex& ix, uint32_t id) {
    const uint2 k = *reinterpret_cast<const uint2*>(&ix.kv[id]);
    return ((uint64_t)k.y << 32) | k.x;
}
__device__ __forceinline__ uint32_t index_find(const RefIndex& ix, uint64_t h) {
    const uint32_t fp = index_fp(h);
    uint32_t b = index_bucket(h, ix.bmask);
    for (;;) {
        const uint4 f = ix.fpb[b];
        const uint32_t id0 = ix.base[b];
        uint32_t m = index_match_mask(f, fp);
        while (m) {
            const uint32_t q = (uint32_t)__ffs((int)m) - 1u;
            m &= m - 1u;
            if (index_key(ix, id0 + q) == h) return id0 + q;
        }
        if (!(f.x & IDX_OVF)) return IDX_NOT_FOUND;   // nothing was ever pushed past this bucket
        b = (b + 1) & ix.bmask;
    }
}

// 64-bit (value, index) wave reductions via shuffles
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

} // namespace rk
