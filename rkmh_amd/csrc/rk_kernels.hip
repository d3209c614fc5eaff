// rk_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the rkmh classify/stream hot path.
//
// What each kernel replaces in the reference (paths relative to /root/reference):
//   k_hash_tiles       mkmh::calc_hashes over arbitrary-length sequences       src/rkmh.cpp:821,831,909,2101
//   k_sort_intersect   mkmh::minhashes (+ mask_by_frequency / minhashes_frequency_filter) and, for
//                      long reads, the intersection loop + argmax               src/rkmh.cpp:822,835,863,916-934
//   (the fused per-read kernel k_classify_tile lives in rk_classify.hip)
//   k_intersect_pair   mkmh::hash_intersection_size for one pair                src/rkmh.cpp:869
//
// Integer hashing work: no MFMA.  The design points are coalesced dword loads of the bases, LDS-staged
// forward + reverse-complement strings, wave ballots for compaction, LDS atomics for the per-reference
// counters and a hash-table index of all reference sketches that stays L2/MALL resident.
#include "rk_kernels.hpp"
#include <cstdlib>

namespace rk {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_to_upper(uint8_t* d, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        signed char c = (signed char)d[i];
        d[i] = (uint8_t)(((int)c - 91) > 0 ? c - 32 : c);
    }
}
hipError_t launch_to_upper(uint8_t* d, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_to_upper, dim3(grid), dim3(256), 0, st, d, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// calc_hashes: one block per tile (grid-stride), tile = <= HASH_TILE_WIN windows of one sequence.
template <int KT>
__global__ __launch_bounds__(256) void k_hash_tiles(const uint8_t* __restrict__ bases, const TileDesc* __restrict__ tiles,
                                                    uint32_t ntiles, uint64_t* __restrict__ out, int32_t* counter,
                                                    uint64_t slots, DevPolicy pol) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[stage_lds_dwords(HASH_TILE_MAXB)];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const TileDesc td = tiles[t];
        if (KT != 0 && td.k != (uint32_t)KT) continue; // other k handled by the generic instance
        if (KT == 0 && td.k == 16) continue;
        __syncthreads(); // previous tile's readers are done
        Staged s = stage_piece(bases, td.base_off, td.nbases, lds, HASH_TILE_MAXB, threadIdx.x, 256,
                               [] { __syncthreads(); });
        for (uint32_t i = threadIdx.x; i < td.nwin; i += 256) {
            uint64_t h = canonical_window<KT>(s, i, (int)td.k, pol);
            out[td.out_off + i] = h;
            if (counter && (pol.counter_counts_zero || h != 0)) atomicAdd(&counter[h % slots], 1);
        }
    }
}
hipError_t launch_hash_tiles(const uint8_t* bases, const TileDesc* tiles, uint32_t ntiles, uint64_t* out,
                             int32_t* counter, uint64_t slots, const DevPolicy& pol, hipStream_t st) {
    if (ntiles == 0) return hipSuccess;
    uint32_t grid = ntiles < 8192 ? ntiles : 8192;
    hipLaunchKernelGGL(k_hash_tiles<16>, dim3(grid), dim3(256), 0, st, bases, tiles, ntiles, out, counter, slots, pol);
    hipLaunchKernelGGL(k_hash_tiles<0>, dim3(grid), dim3(256), 0, st, bases, tiles, ntiles, out, counter, slots, pol);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// shared tail of both classify kernels: per-reference counters in LDS -> argmax/diff of rkmh.cpp:874-883
// (first index attaining the max wins; diff = max - max(earlier entries), -1 when there is none).
__device__ __forceinline__ void wave_argmax_diff(const int* sh, int R, int lane, int& max_id, int& max_shared, int& diff) {
    int best = -1, besti = 0x7fffffff;
    for (int j = lane; j < R; j += 64) {
        int v = sh[j];
        if (v > best) { best = v; besti = j; }
    }
    int gmax = wave_max_i32(best);
    int gid = wave_min_i32(best == gmax ? besti : 0x7fffffff);
    int prev = -1;
    for (int j = lane; j < gid; j += 64) { int v = sh[j]; prev = v > prev ? v : prev; }
    prev = wave_max_i32(prev);
    max_id = gid; max_shared = gmax; diff = gmax - prev;
}

__device__ __forceinline__ void accumulate_posting(const RefIndex& ix, uint32_t slot, uint32_t rank, int* sh) {
    uint32_t v = ix.kv[slot].z;
    if (!(v >> 31)) {
        if (((v >> 29) & 3u) == 0u) {
            uint32_t ref = v & 0xFFFFFu, mult = (v >> 20) & 0x1FFu;
            if (rank < mult) atomicAdd(&sh[ref], 1);
        } else if (rank == 0) {
            atomicAdd(&sh[v & 0x7FFu], 1);
            atomicAdd(&sh[(v >> 11) & 0x7FFu], 1);
        }
    } else {
        uint32_t off = v & 0x7fffffffu;
        uint32_t cnt = ix.post[off];
        for (uint32_t c = 0; c < cnt; ++c) {
            uint32_t ref = ix.post[off + 1 + 2 * c], mult = ix.post[off + 2 + 2 * c];
            if (rank < mult) atomicAdd(&sh[ref], 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// minhashes (+filters) and optional intersection: one block per sequence, bitonic sort in LDS.
__global__ void k_sort_intersect(SortArgs a, RefIndex ix, int has_ix, DevPolicy pol) {
    extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
    uint64_t* v = sm;
    // [nref] counters (in LDS, or this block's row of a.gcount for panels too large for it), then one int: number of zeros
    int* lsh = reinterpret_cast<int*>(sm + a.P);
    const int nl = (has_ix && !a.gcount) ? ix.nref : 0;
    int* sh = a.gcount ? a.gcount + (size_t)blockIdx.x * (size_t)ix.nref : lsh;
    int& s_nz = lsh[nl];
    // pre-selection scratch (only when a.preselect): histogram, threshold bucket, a few scalars
    uint32_t* hist = reinterpret_cast<uint32_t*>(lsh + nl + 4);
    uint64_t* side = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(hist + (1 << PRESEL_BITS) + 8 + 1024 + 64) + 7) & ~(uintptr_t)7);
    uint32_t* ps = hist + (1 << PRESEL_BITS);            // [0] taken so far [1] side count [2] bin [3] below [4] bucket count
    uint32_t* csum = ps + 8;                             // [<= 1024] per-thread partial sums, then [64] per-lane sums
    const int tid = threadIdx.x, T = blockDim.x;
    const uint32_t P = a.P;
    auto filtered = [&](uint64_t h) -> uint64_t {
        if (a.filter_mode == FILTER_MASK_MIN) {            // mask_by_frequency, rkmh.cpp:916
            int c = a.counter[h % a.slots];
            if (pol.mask_strict_less ? (c < a.fmin) : (c <= a.fmin)) h = 0;
        } else if (a.filter_mode == FILTER_RANGE && h != 0) { // minhashes_frequency_filter, rkmh.cpp:835
            int c = a.counter[h % a.slots];
            bool keep = pol.freq_max_inclusive ? (c >= a.fmin && c <= a.fmax) : (c >= a.fmin && c < a.fmax);
            if (!keep) h = 0;
        } else if (a.filter_mode == FILTER_KEYMASK && h != 0 && has_ix && ix.keepkey) { // the mask through the index keys (no selection follows)
            const uint32_t slot = index_find(ix, h);
            if (slot != IDX_NOT_FOUND && !((ix.keepkey[slot >> 5] >> (slot & 31u)) & 1u)) h = 0;
        }
        return h;
    };
    for (uint32_t li = blockIdx.x; li < a.nlist; li += gridDim.x) {
        const uint32_t id = a.seq_ids[li];
        const bool sel = a.sel_len != nullptr;
        const uint64_t seg = sel ? 0 : a.seg_off[id];
        const uint32_t n = sel ? a.sel_len[0] : (uint32_t)(a.seg_off[id + 1] - seg);
        const uint64_t* src = sel ? a.sel_hashes : a.hashes;
        uint32_t n_sort = n; // entries of v that belong to the sequence
        __syncthreads();
        if (a.preselect && n > P) {
            // ---- exact bottom-S of the non-zero (filtered) hashes by most-significant-digit radix select ----
            // invariant: every hash < lo is taken (there are S - need of them); the answer's remaining `need` elements
            // are the smallest ones of the bucket [lo, lo + 2^shift)
            uint64_t lo = 0;
            uint32_t shift = 64, need = (uint32_t)a.S;
            bool take_all = false;
            for (;;) {
                const uint32_t bits = shift >= (uint32_t)PRESEL_BITS ? (uint32_t)PRESEL_BITS : shift;
                const uint32_t nbins = 1u << bits;
                const uint32_t dsh = shift - bits;
                for (uint32_t b = tid; b < nbins; b += T) hist[b] = 0;
                __syncthreads();
                for (uint32_t t = tid; t < n; t += T) {
                    const uint64_t h = filtered(src[seg + t]);
                    if (h != 0 && (shift == 64 || (h >> shift) == (lo >> shift))) atomicAdd(&hist[(uint32_t)(h >> dsh) & (nbins - 1)], 1u);
                }
                __syncthreads();
                // bin where the running count reaches `need`: per-thread chunk sums, then one wave scans the chunks
                const uint32_t per = (nbins + (uint32_t)T - 1) / (uint32_t)T;
                {
                    uint32_t sum = 0;
                    for (uint32_t j = 0; j < per; ++j) { const uint32_t b = (uint32_t)tid * per + j; if (b < nbins) sum += hist[b]; }
                    csum[tid] = sum;
                }
                __syncthreads();
                // two-level scan: 64 lanes sum T/64 chunk sums each, lane 0 walks the 64 lane sums, then one lane's
                // chunks, then one chunk's bins -- a few dozen dependent steps instead of thousands
                const uint32_t cpl = ((uint32_t)T + 63u) / 64u; // chunks per lane
                if (tid < 64) {
                    uint32_t sum = 0;
                    for (uint32_t j = 0; j < cpl; ++j) { const uint32_t c = (uint32_t)tid * cpl + j; if (c < (uint32_t)T) sum += csum[c]; }
                    csum[1024 + tid] = sum;
                }
                __syncthreads();
                if (tid == 0) {
                    uint32_t acc = 0, lane_ = 0;
                    for (; lane_ < 64; ++lane_) { if (acc + csum[1024 + lane_] >= need) break; acc += csum[1024 + lane_]; }
                    if (lane_ == 64) { ps[2] = nbins; ps[3] = acc; ps[4] = 0; } // fewer than `need` left: take everything
                    else {
                        uint32_t chunk = lane_ * cpl;
                        for (;; ++chunk) { if (acc + csum[chunk] >= need) break; acc += csum[chunk]; }
                        uint32_t b = chunk * per;
                        for (;; ++b) { if (acc + hist[b] >= need) break; acc += hist[b]; }
                        ps[2] = b; ps[3] = acc; ps[4] = hist[b];
                    }
                }
                __syncthreads();
                const uint32_t bin = ps[2], below = ps[3], bcount = ps[4];
                __syncthreads();
                if (bin == nbins) { take_all = true; break; }
                need -= below;
                lo |= (uint64_t)bin << dsh;
                shift = dsh;
                if (shift == 0 || bcount <= (uint32_t)PRESEL_SIDE) break;
            }
            // collect: everything below the bucket, and the bucket itself on the side
            if (tid == 0) { ps[0] = 0; ps[1] = 0; }
            __syncthreads();
            for (uint32_t t = tid; t < n; t += T) {
                const uint64_t h = filtered(src[seg + t]);
                if (h == 0) continue;
                if (take_all || h < lo) { const uint32_t pos = atomicAdd(&ps[0], 1u); if (pos < P) v[pos] = h; }
                else if (shift == 0 ? h == lo : (h >> shift) == (lo >> shift)) {
                    const uint32_t pos = atomicAdd(&ps[1], 1u);
                    if (pos < (uint32_t)PRESEL_SIDE) side[pos] = h;
                }
            }
            __syncthreads();
            uint32_t taken = ps[0];
            if (!take_all) {
                if (shift == 0) { // the bucket is one value repeated: `need` copies of it
                    for (uint32_t t = tid; t < need; t += T) v[taken + t] = lo;
                } else { // <= PRESEL_SIDE elements: rank sort, the `need` smallest complete the selection
                    const uint32_t ns = ps[1];
                    for (uint32_t i = tid; i < ns; i += T) {
                        const uint64_t x = side[i];
                        uint32_t r = 0;
                        for (uint32_t j = 0; j < ns; ++j) { const uint64_t y = side[j]; r += (y < x || (y == x && j < i)) ? 1u : 0u; }
                        if (r < need) v[taken + r] = x;
                    }
                }
                taken += need;
            }
            __syncthreads();
            for (uint32_t t = taken + tid; t < P; t += T) v[t] = ~0ull;
            n_sort = taken;
        } else
        for (uint32_t t = tid; t < P; t += T) {
            uint64_t h = ~0ull;
            if (t < n) {
                h = src[seg + t];
                if (!sel) h = filtered(h);
            }
            v[t] = h;
        }
        if (tid == 0) s_nz = 0;
        __syncthreads();
        for (uint32_t size = 2; size <= P; size <<= 1) {
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = tid; t < (P >> 1); t += T) {
                    uint32_t lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
                    uint32_t hi = lo | stride;
                    bool asc = ((lo & size) == 0);
                    uint64_t x = v[lo], y = v[hi];
                    if ((x > y) == asc) { v[lo] = y; v[hi] = x; }
                }
                __syncthreads();
            }
        }
        // zeros sort first: nz = number of zero entries among the n real ones
        for (uint32_t t = tid; t < n_sort; t += T)
            if (v[t] == 0 && (t + 1 == n_sort || v[t + 1] != 0)) s_nz = (int)(t + 1);
        __syncthreads();
        const uint32_t nz = (uint32_t)s_nz;
        const uint32_t m = (n_sort - nz) < (uint32_t)a.S ? (n_sort - nz) : (uint32_t)a.S;
        if (a.write_back)
            for (uint32_t t = tid; t < n; t += T) a.hashes[seg + t] = v[t];
        if (a.sketches) {
            uint64_t* sk = a.sketches + (uint64_t)id * (uint64_t)a.S;
            for (uint32_t t = tid; t < (uint32_t)a.S; t += T) sk[t] = t < m ? v[nz + t] : 0ull;
        }
        if (a.lens && tid == 0) a.lens[id] = (int32_t)m;
        if (has_ix && a.out4) {
            for (int j = tid; j < ix.nref; j += T) sh[j] = 0;
            __syncthreads();
            for (uint32_t t = tid; t < m; t += T) {
                uint64_t h = v[nz + t];
                uint32_t u = t;
                while (u > 0 && v[nz + u - 1] == h) --u;   // rank among equal sketch entries (multiset merge)
                uint32_t slot = index_find(ix, h);
                if (slot != IDX_NOT_FOUND) accumulate_posting(ix, slot, t - u, sh);
            }
            __syncthreads();
            const int R_arg = a.argmax_n > 0 ? a.argmax_n : ix.nref;
            if (tid < 64) {
                int mi, ms, df;
                wave_argmax_diff(sh, R_arg, tid, mi, ms, df);
                if (tid == 0) {
                    int4 r = make_int4(mi, ms, df, (int)m);
                    reinterpret_cast<int4*>(a.out4)[id] = r;
                }
            }
            if (a.tail_counts) { // raw counts of the references that take no part in the argmax (lineage / sublineage sets)
                const int nt = ix.nref - R_arg;
                for (int j = tid; j < nt; j += T) a.tail_counts[(size_t)id * (size_t)nt + j] = sh[R_arg + j];
            }
        }
    }
}
static size_t sort_intersect_lds(uint32_t P, size_t counters, bool preselect) {
    size_t lds = (size_t)P * 8 + counters * 4 + 16; // sort buffer + counters + s_nz
    if (preselect) lds += ((size_t)(1 << PRESEL_BITS) + 8 + 1024 + 64) * 4 + (size_t)PRESEL_SIDE * 8 + 16;
    return lds;
}
constexpr size_t LDS_PER_WORKGROUP = 160 * 1024; // gfx950
uint32_t sort_intersect_global_rows(uint32_t P, int nref) {
    if (nref <= 0 || sort_intersect_lds(P, (size_t)nref, true) <= LDS_PER_WORKGROUP) return 0;
    const size_t rows = ((size_t)256 << 20) / ((size_t)nref * 4); // <= 256 MB of counter rows
    return (uint32_t)(rows < 64 ? 64 : (rows > 4096 ? 4096 : rows));
}
hipError_t launch_sort_intersect(const SortArgs& a, const RefIndex* ix, const DevPolicy& pol, hipStream_t st) {
    if (a.nlist == 0) return hipSuccess;
    int T = (int)(a.P / 2);
    if (T < 64) T = 64;
    if (T > 1024) T = 1024;
    RefIndex z{};
    const RefIndex& use = ix ? *ix : z;
    const size_t lds = sort_intersect_lds(a.P, (ix && !a.gcount) ? (size_t)ix->nref : 0, a.preselect != 0);
    if (lds > LDS_PER_WORKGROUP) return hipErrorInvalidValue; // callers size SortArgs::gcount with sort_intersect_global_rows
    uint32_t grid = a.nlist < 65535 ? a.nlist : 65535;
    if (a.gcount && grid > a.gcount_rows) grid = a.gcount_rows;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_intersect),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_sort_intersect, dim3(grid), dim3(T), lds, st, a, use, ix ? 1 : 0, pol);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// mask_by_frequency (rkmh.cpp:916) keeps a hash when its slot's count passes the threshold.  Once the table is final that
// comparison is a property of the SLOT: one bit per slot (bit = keep), built in one streaming pass over the table.
__global__ __launch_bounds__(256) void k_keep_bits(const int32_t* __restrict__ counter, uint64_t slots, int min_occ, int strict_less,
                                                   uint32_t* __restrict__ bits, uint64_t nwords) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        const uint64_t s0 = w * 32;
#pragma unroll 8
        for (uint32_t b = 0; b < 32; ++b) {
            const uint64_t sidx = s0 + b;
            if (sidx < slots) {
                const int c = counter[sidx];
                const bool masked = strict_less ? (c < min_occ) : (c <= min_occ);
                v |= (masked ? 0u : 1u) << b;
            }
        }
        bits[w] = v;
    }
}
hipError_t launch_keep_bits(const int32_t* counter, uint64_t slots, int min_occ, const DevPolicy& pol, uint32_t* bits, hipStream_t st) {
    const uint64_t nwords = (slots + 31) / 32;
    uint64_t blocks = (nwords + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_keep_bits, dim3((uint32_t)blocks), dim3(256), 0, st, counter, slots, min_occ, pol.mask_strict_less, bits, nwords);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// -M with a bounded min_num (rk_set_min_num_bound; rkmh.cpp:916-917, :938).  mask_by_frequency changes what a read shares with
// the references only through the windows whose hash is a KEY of the index, and whether a key survives is a property of the key:
// one keep bit per key id, taken from the depth map once (full table: counter[key % slots]; compact table: counter[sid[key id]]).
__global__ __launch_bounds__(256) void k_keep_keys(const uint4* __restrict__ kv, uint32_t nkeys, const int32_t* __restrict__ counter, uint64_t slots,
                                                   const uint32_t* __restrict__ key_sid, int min_occ, int strict_less, uint32_t* __restrict__ bits) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 32u >= nkeys) return;
    uint32_t v = 0;
    for (uint32_t b = 0; b < 32; ++b) {
        const uint32_t j = w * 32u + b;
        if (j >= nkeys) break;
        int c;
        if (key_sid) c = counter[key_sid[j]];
        else {
            const uint2 k = *reinterpret_cast<const uint2*>(&kv[j]);
            c = counter[(((uint64_t)k.y << 32) | k.x) % slots];
        }
        const bool masked = strict_less ? (c < min_occ) : (c <= min_occ);
        v |= (masked ? 0u : 1u) << b;
    }
    bits[w] = v;
}
hipError_t launch_keep_keys(const RefIndex& ix, uint32_t nkeys, const int32_t* counter, uint64_t slots, const uint32_t* key_sid, int min_occ,
                            const DevPolicy& pol, uint32_t* bits, hipStream_t st) {
    const uint32_t nwords = (nkeys + 31u) / 32u;
    if (!nwords) return hipSuccess;
    hipLaunchKernelGGL(k_keep_keys, dim3((nwords + 255u) / 256u), dim3(256), 0, st, ix.kv, nkeys, counter, slots, key_sid, min_occ, pol.mask_strict_less, bits);
    return hipGetLastError();
}
// the exact k-mer map with the dropped keys turned into zero-hash k-mers (a masked hash IS 0, rkmh.cpp:916): cells[i] = (cell of
// found k-mer i, its key id or IDX_NOT_FOUND); km1m is a copy of the map
__global__ __launch_bounds__(256) void k_km1_mask(const uint2* __restrict__ cells, uint32_t n, const uint32_t* __restrict__ keepkey, uint32_t* __restrict__ km1m,
                                                  uint32_t vmask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 e = cells[i];
    if (e.y == IDX_NOT_FOUND) return;
    if (!((keepkey[e.y >> 5] >> (e.y & 31u)) & 1u)) km1m[e.x] = (km1m[e.x] & ~vmask) | (vmask - 1u);
}
// the key array of the hash-space kernels with the mask folded in: kvm[j].w = 1 for a key the mask drops (the drain loads kv[j]
// anyway to verify a hit: no further load for the keep bit)
__global__ __launch_bounds__(256) void k_kv_mask(const uint4* __restrict__ kv, uint32_t nkeys, const uint32_t* __restrict__ keepkey, uint4* __restrict__ kvm) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nkeys) return; // (the array holds nkeys + 1 entries)
    uint4 e = kv[j];
    e.w = (j < nkeys && !((keepkey[j >> 5] >> (j & 31u)) & 1u)) ? 1u : 0u;
    kvm[j] = e;
}
hipError_t launch_kv_mask(const uint4* kv, uint32_t nkeys, const uint32_t* keepkey, uint4* kvm, hipStream_t st) {
    hipLaunchKernelGGL(k_kv_mask, dim3((nkeys + 256u) / 256u), dim3(256), 0, st, kv, nkeys, keepkey, kvm);
    return hipGetLastError();
}
hipError_t launch_km1_mask(const uint2* cells, uint32_t n, const uint32_t* keepkey, uint32_t* km1m, uint32_t vmask, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_km1_mask, dim3((n + 255u) / 256u), dim3(256), 0, st, cells, n, keepkey, km1m, vmask);
    return hipGetLastError();
}
// min(min_num, bound) for the rows the fused kernels answered (not -2): num_mins only ever meets `num_mins <= min_matches`
// (rkmh.cpp:938), so no caller needs more.  A row whose max_shared already reaches the bound needs nothing (max_shared <=
// num_mins); for the others G lanes take one read and hash its windows G at a time (LDS image of the piece, both strands, as
// k_hash_tiles does) until `bound` of them survive the mask -- a read whose first windows survive costs G hashes and G bits of
// the slot bitmap instead of all of them.
constexpr int PROBE_MAXB = 32 + MAX_K;
template <int G>
__global__ __launch_bounds__(256) void k_min_num_probe(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ offs, uint32_t nreads, KsArr ks,
                                                       int S, int bound, const uint32_t* __restrict__ keepbits, uint64_t slots, uint64_t slots_m,
                                                       DevPolicy pol, int32_t* __restrict__ out4) {
    constexpr int GROUPS = 256 / G;
    __shared__ uint32_t lds[GROUPS * stage_lds_dwords(PROBE_MAXB)];
    const int tid = threadIdx.x % G, grp = threadIdx.x / G, lane = threadIdx.x & 63;
    uint32_t* my = lds + grp * stage_lds_dwords(PROBE_MAXB);
    const uint32_t i = blockIdx.x * GROUPS + grp;
    const int want = bound < S ? bound : S;
    bool done = i >= nreads;
    uint32_t o = 0, len = 0;
    int cnt = 0;
    if (!done) {
        const int2 r = *reinterpret_cast<const int2*>(out4 + 4 * (size_t)i);
        if (r.x == -2) done = true; // the general path answers this read (exactly)
        else if (r.y >= want) { cnt = want; done = true; out4[4 * (size_t)i + 3] = want; }
        else { o = offs[i]; len = offs[i + 1] - o; }
    }
    const bool mine = !done;
    int j = 0;
    uint32_t w0 = 0;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    while (__ballot(!done)) { // (the lanes of a group always agree; a wave holds 64 / G groups)
        bool go = !done;
        int k = 1;
        uint32_t nw = 0;
        if (go) {
            k = ks.k[j]; nw = (uint32_t)num_windows((int)len, k, pol.drop_last_window);
            if (w0 >= nw) { // this size is exhausted: on to the next one (its windows start with the wave's next step)
                ++j; w0 = 0; go = false;
                if (j >= ks.n) done = true;
            }
        }
        const uint32_t nwin = go ? ((nw - w0) < (uint32_t)G ? (nw - w0) : (uint32_t)G) : 0u;
        const Staged st = stage_piece(bases, (uint64_t)o + w0, go ? nwin + (uint32_t)k - 1u : 0u, my, PROBE_MAXB, tid, G, sync);
        bool surv = false;
        if (go && (uint32_t)tid < nwin) {
            const uint64_t h = canonical_window<0>(st, (uint32_t)tid, k, pol);
            if (h != 0) {
                const uint64_t sl = mod_slots(h, slots, slots_m);
                surv = ((keepbits[sl >> 5] >> ((uint32_t)sl & 31u)) & 1u) != 0u;
            }
        }
        const uint64_t m = __ballot(surv);
        if (go) {
            cnt += __popcll((m >> (lane & ~(G - 1))) & (G == 64 ? ~0ull : ((1ull << G) - 1ull)));
            w0 += (uint32_t)G;
            if (cnt >= want) done = true;
        }
        sync(); // the image is rebuilt by the next step
    }
    if (mine && tid == 0) out4[4 * (size_t)i + 3] = cnt < want ? cnt : want;
}
hipError_t launch_min_num_probe(const uint8_t* bases, const uint32_t* offs, uint32_t nreads, const KsArr& ks, int S, int bound,
                                const uint32_t* keepbits, uint64_t slots, const DevPolicy& pol, int32_t* out4, hipStream_t st) {
    if (!nreads) return hipSuccess;
    const uint64_t slots_m = slots ? ~0ull / slots : 0;
    const int want = bound < S ? bound : S;
#define RK_PROBE(G) hipLaunchKernelGGL((k_min_num_probe<G>), dim3((nreads + (256 / G) - 1) / (256 / G)), dim3(256), 0, st, bases, offs, nreads, ks, S, bound, \
                                       keepbits, slots, slots_m, pol, out4)
    if (want <= 6) RK_PROBE(8);
    else if (want <= 14) RK_PROBE(16);
    else RK_PROBE(32);
#undef RK_PROBE
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Exhaustive enumeration behind the k-mer-space kernel (rk_kmer.hip): EVERY k-mer of the 4^k universe (k <= 16) is hashed
// exactly as calc_hashes would hash it as a window; those whose canonical hash is a key of the index -- the true preimages of
// the sketch hashes and any other k-mer that happens to collide with one -- and those hashing to 0 are listed.  The host builds
// the group filter (kf4) and the exact map (km1) from the list, so that kernel's decision "this window cannot matter" is a
// theorem about this index, not a probabilistic filter.  4^16 k-mers take ~25 ms on MI355X (two murmurs + one 4-byte read of
// the hash-space filter each; the bucket table is only touched by the few that pass).
// (Only the smaller member of a strand pair is hashed -- half of the k-mers a wave walks over -- so each wave queues the ones it
// keeps in LDS and hashes them 64 at a time: all lanes busy in the part that costs, instead of every second one.)
template <typename V>
struct EnumQueue {
    V* q;        // this wave's 128 entries
    uint32_t n;  // queued (wave-uniform)
    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // every lane offers one k-mer (keep: whether it is taken); afterwards full(): 64 or more are queued
    __device__ __forceinline__ void push(bool keep, V v, int lane) {
        const uint64_t m = __ballot(keep);
        if (keep) q[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
        n += (uint32_t)__popcll(m);
        sync();
    }
    // the first min(n, 64) entries, one per lane (has: this lane got one); the rest move to the front
    __device__ __forceinline__ V pop(int lane, bool& has) {
        const uint32_t take = n < 64u ? n : 64u, rest = n - take;
        has = (uint32_t)lane < take;
        const V v = q[lane];                 // (lanes past the end read stale entries they do not use)
        const V w = q[64 + lane];
        sync();
        if ((uint32_t)lane < rest) q[lane] = w;
        n = rest;
        sync();
        return v;
    }
};

__global__ __launch_bounds__(256) void k_enum_kmers(RefIndex ix, DevPolicy pol, int k, uint64_t total, uint32_t* stats, uint2* list,
                                                    uint32_t list_cap) {
    __shared__ uint32_t qmem[4][128];
    const int lane = threadIdx.x & 63;
    EnumQueue<uint32_t> Q{qmem[threadIdx.x >> 6], 0u};
    auto hash_one = [&](uint32_t v) {
        const uint64_t h = canonical_packed(v, k, pol.seed, pol.fold);
        uint32_t slot = IDX_NOT_FOUND;
        if (h != 0) {
            if (ix.pre) { // hash-space filter first: one word instead of a bucket for the 99.99 % that are not keys
                const uint32_t bm = index_pre_bits(h);
                if ((ix.pre[index_pre_word(h, ix.pmask)] & bm) != bm) return;
            }
            slot = index_find(ix, h);
            if (slot == IDX_NOT_FOUND) return;
        }
        const uint32_t pos = atomicAdd(stats, 1u);
        if (pos < list_cap) list[pos] = make_uint2(v, slot); // slot = IDX_NOT_FOUND: the k-mer hashes to 0
    };
    // (the loop bound is the same for all lanes of a wave: v64 - lane is)
    for (uint64_t v64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v64 - (uint64_t)lane < total; v64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)v64;
        Q.push(v64 < total && packed_revcomp(v, k) >= v, v, lane); // each strand pair once, through its smaller member (a palindrome is its own)
        if (Q.n >= 64u) { bool has; const uint32_t x = Q.pop(lane, has); if (has) hash_one(x); }
    }
    while (Q.n) { bool has; const uint32_t x = Q.pop(lane, has); if (has) hash_one(x); }
}
// the same for wide k-mers (16 < k <= KW_MAX_K): 4^k is 1.7 * 10^10 at k = 17 and 1.1 * 10^12 at k = 20 -- 0.1 s to several seconds, once per
// reference set (rk_set_kmer_cache keeps the list); list entries (k-mer low, k-mer high, key id or IDX_NOT_FOUND, 0)
__global__ __launch_bounds__(256) void k_enum_kmers64(RefIndex ix, DevPolicy pol, int k, uint64_t total, uint32_t* stats, uint4* list, uint32_t list_cap) {
    __shared__ uint64_t qmem[4][128];
    const int lane = threadIdx.x & 63;
    EnumQueue<uint64_t> Q{qmem[threadIdx.x >> 6], 0u};
    auto hash_one = [&](uint64_t v) {
        const uint64_t h = canonical_packed64(v, k, pol.seed, pol.fold);
        uint32_t slot = IDX_NOT_FOUND;
        if (h != 0) {
            if (ix.pre) {
                const uint32_t bm = index_pre_bits(h);
                if ((ix.pre[index_pre_word(h, ix.pmask)] & bm) != bm) return;
            }
            slot = index_find(ix, h);
            if (slot == IDX_NOT_FOUND) return;
        }
        const uint32_t pos = atomicAdd(stats, 1u);
        if (pos < list_cap) list[pos] = make_uint4((uint32_t)v, (uint32_t)(v >> 32), slot, 0u);
    };
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v - (uint64_t)lane < total; v += (uint64_t)gridDim.x * blockDim.x) {
        Q.push(v < total && packed_revcomp64(v, k) >= v, v, lane);
        if (Q.n >= 64u) { bool has; const uint64_t x = Q.pop(lane, has); if (has) hash_one(x); }
    }
    while (Q.n) { bool has; const uint64_t x = Q.pop(lane, has); if (has) hash_one(x); }
}
hipError_t launch_enum_kmers(const RefIndex& ix, const DevPolicy& pol, int k, uint32_t* stats, uint2* list, uint32_t list_cap,
                             hipStream_t st) {
    const uint64_t total = 1ull << (2 * k);
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (k > 16) hipLaunchKernelGGL(k_enum_kmers64, dim3((uint32_t)blocks), dim3(256), 0, st, ix, pol, k, total, stats, reinterpret_cast<uint4*>(list), list_cap);
    else hipLaunchKernelGGL(k_enum_kmers, dim3((uint32_t)blocks), dim3(256), 0, st, ix, pol, k, total, stats, list, list_cap);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Radix select for sequences with more hashes than the LDS sorter holds (chromosome-scale references).
// state dwords: [0,1] prefix (u64, high bits decided so far)  [2] remaining rank inside the prefix bucket
//               [3] count strictly below the prefix  [4] take_all flag  [5] compaction cursor  [8] candidates out
constexpr int SEL_BITS = 13;
constexpr int SEL_BINS = 1 << SEL_BITS;
__device__ __forceinline__ uint64_t sel_filter(uint64_t h, const int32_t* counter, uint64_t slots, int mode, int fmin, int fmax,
                                               const DevPolicy& pol) {
    if (mode == FILTER_MASK_MIN) {
        int c = counter[h % slots];
        if (pol.mask_strict_less ? (c < fmin) : (c <= fmin)) h = 0;
    } else if (mode == FILTER_RANGE && h != 0) {
        int c = counter[h % slots];
        bool keep = pol.freq_max_inclusive ? (c >= fmin && c <= fmax) : (c >= fmin && c < fmax);
        if (!keep) h = 0;
    }
    return h;
}
// digit d (0..4) covers bits [shift, shift+width): 13,13,13,13,12 bits from the top
__device__ __forceinline__ void sel_digit(int d, int& shift, int& width) {
    width = d < 4 ? SEL_BITS : 64 - 4 * SEL_BITS;
    shift = d < 4 ? 64 - SEL_BITS * (d + 1) : 0;
}
__global__ __launch_bounds__(256) void k_sel_hist(const uint64_t* __restrict__ h, uint64_t n, const int32_t* counter, uint64_t slots,
                                                  int mode, int fmin, int fmax, DevPolicy pol, const uint32_t* state, uint32_t* hist,
                                                  int d) {
    __shared__ uint32_t lh[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    int shift, width;
    sel_digit(d, shift, width);
    const uint64_t prefix = ((uint64_t)state[1] << 32) | state[0];
    const bool take_all = state[4] != 0;
    if (!take_all) {
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
            const uint64_t v = sel_filter(h[i], counter, slots, mode, fmin, fmax, pol);
            if (v == 0) continue;
            if (d > 0 && (v >> (shift + width)) != prefix) continue;
            atomicAdd(&lh[(uint32_t)(v >> shift) & ((1u << width) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
__global__ __launch_bounds__(1024) void k_sel_scan(uint32_t* state, uint32_t* hist, int d, int S) {
    __shared__ uint32_t part[1024];
    int shift, width;
    sel_digit(d, shift, width);
    const int bins = 1 << width, per = SEL_BINS / 1024; // 8 bins per thread
    const int tid = threadIdx.x;
    if (d == 0 && tid == 0) state[2] = (uint32_t)S; // everything else was zeroed by the launcher
    __syncthreads();
    uint32_t loc[8], sum = 0;
    for (int j = 0; j < per; ++j) { int b = tid * per + j; loc[j] = b < bins ? hist[b] : 0; sum += loc[j]; }
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { // inclusive scan of the per-thread sums
        uint32_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const uint32_t total = part[1023];
    const uint32_t remaining = state[2];
    __syncthreads();
    if (state[4] == 0) {
        if (d == 0 && total < remaining) { if (tid == 0) state[4] = 1; } // fewer kept hashes than S: take them all
        else {
            uint32_t before = part[tid] - sum; // kept hashes in bins before this thread's
            for (int j = 0; j < per; ++j) {
                if (before < remaining && before + loc[j] >= remaining) { // the S-th smallest lies in this bin
                    const uint64_t prefix = (((uint64_t)state[1] << 32) | state[0]);
                    const uint64_t np = d == 0 ? (uint64_t)(tid * per + j) : ((prefix << width) | (uint64_t)(tid * per + j));
                    state[0] = (uint32_t)np; state[1] = (uint32_t)(np >> 32);
                    state[2] = remaining - before;
                    state[3] += before;
                }
                before += loc[j];
            }
        }
    }
    for (int j = 0; j < per; ++j) hist[tid * per + j] = 0; // ready for the next digit
}
__global__ __launch_bounds__(256) void k_sel_compact(const uint64_t* __restrict__ h, uint64_t n, const int32_t* counter, uint64_t slots,
                                                     int mode, int fmin, int fmax, DevPolicy pol, uint32_t* state, uint64_t* out) {
    const uint64_t T = ((uint64_t)state[1] << 32) | state[0]; // the S-th smallest kept hash
    const bool take_all = state[4] != 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = sel_filter(h[i], counter, slots, mode, fmin, fmax, pol);
        if (v != 0 && (take_all || v < T)) out[atomicAdd(&state[5], 1u)] = v;
    }
    if (blockIdx.x == 0 && !take_all) { // state[2] copies of the threshold value complete the bottom-S multiset
        const uint32_t base = state[3], copies = state[2];
        for (uint32_t j = threadIdx.x; j < copies; j += 256) out[base + j] = T;
    }
}
__global__ void k_sel_finish(uint32_t* state) { state[8] = state[4] ? state[5] : state[3] + state[2]; }

hipError_t launch_select_bottom(const uint64_t* hashes, uint64_t n, int S, const int32_t* counter, uint64_t slots,
                                int filter_mode, int fmin, int fmax, const DevPolicy& pol, uint32_t* sel_state,
                                uint32_t* hist, uint64_t* sel_out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(hist, 0, SEL_BINS * 4, st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(sel_state, 0, 16 * 4, st); // the first histogram pass already reads the take-all flag
    if (e != hipSuccess) return e;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    for (int d = 0; d < 5; ++d) {
        hipLaunchKernelGGL(k_sel_hist, dim3(grid), dim3(256), 0, st, hashes, n, counter, slots, filter_mode, fmin, fmax, pol,
                           (const uint32_t*)sel_state, hist, d);
        hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(1024), 0, st, sel_state, hist, d, S);
    }
    // the compaction writes values < T through the cursor (which ends at state[3]) and the copies of T behind them
    hipLaunchKernelGGL(k_sel_compact, dim3(grid), dim3(256), 0, st, hashes, n, counter, slots, filter_mode, fmin, fmax, pol,
                       sel_state, sel_out);
    hipLaunchKernelGGL(k_sel_finish, dim3(1), dim3(1), 0, st, sel_state);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// hash_intersection_size for one pair of ascending arrays (leading zeros skipped, both advance on
// equality => sum over values of min(mult_a, mult_b)).  One block, binary searches into b.
__global__ __launch_bounds__(256) void k_intersect_pair(const uint64_t* __restrict__ a, int na,
                                                        const uint64_t* __restrict__ b, int nb, int* out) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int local = 0;
    for (int t = threadIdx.x; t < na; t += 256) {
        uint64_t h = a[t];
        if (h == 0) continue;
        int u = t;
        while (u > 0 && a[u - 1] == h) --u;
        int rank = t - u;
        int lo = 0, hi = nb;                 // lower_bound
        while (lo < hi) { int mid = (lo + hi) >> 1; if (b[mid] < h) lo = mid + 1; else hi = mid; }
        int first = lo;
        hi = nb;                             // upper_bound
        while (lo < hi) { int mid = (lo + hi) >> 1; if (b[mid] <= h) lo = mid + 1; else hi = mid; }
        if (rank < lo - first) ++local;
    }
    atomicAdd(&total, local);
    __syncthreads();
    if (threadIdx.x == 0) *out = total;
}
hipError_t launch_intersect_pair(const uint64_t* a, int na, const uint64_t* b, int nb, int* out, hipStream_t st) {
    hipLaunchKernelGGL(k_intersect_pair, dim3(1), dim3(256), 0, st, a, na, b, nb, out);
    return hipGetLastError();
}

// mkmh::hash_intersection (7 arguments, equiv.hpp:308,340,364): the matches themselves, ascending, at most `cap` of them.
// Same multiset rule as k_intersect_pair (occurrence `rank` of a hash in a matches while b holds more than `rank` copies);
// the block compacts 256 elements of a per round with wave ballots.
__global__ __launch_bounds__(256) void k_intersect_pair_emit(const uint64_t* __restrict__ a, int na, const uint64_t* __restrict__ b,
                                                             int nb, int cap, uint64_t* __restrict__ out, int* n_out) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int running = 0;
    for (int base = 0; base < na; base += 256) {
        const int t = base + (int)threadIdx.x;
        bool hit = false;
        uint64_t h = 0;
        if (t < na && (h = a[t]) != 0) {
            int u = t;
            while (u > 0 && a[u - 1] == h) --u;
            const int rank = t - u;
            int lo = 0, hi = nb;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (b[mid] < h) lo = mid + 1; else hi = mid; }
            const int first = lo;
            hi = nb;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (b[mid] <= h) lo = mid + 1; else hi = mid; }
            hit = rank < lo - first;
        }
        const uint64_t m = __ballot(hit);
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int before = running;
        for (int j = 0; j < w; ++j) before += wsum[j];
        const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && pos < cap) out[pos] = h;
        running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = running < cap ? running : cap;
}
hipError_t launch_intersect_pair_emit(const uint64_t* a, int na, const uint64_t* b, int nb, int cap, uint64_t* out, int* n_out, hipStream_t st) {
    hipLaunchKernelGGL(k_intersect_pair_emit, dim3(1), dim3(256), 0, st, a, na, b, nb, cap, out, n_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// marks every read "reroute through the general path" (used when the fused kernel cannot take the batch)
__global__ __launch_bounds__(256) void k_fill_reroute(int32_t* out4, uint32_t nreads) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < nreads) reinterpret_cast<int4*>(out4)[i] = make_int4(-2, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_scatter_rows(const int32_t* rows, const uint32_t* ids, uint32_t m, int32_t* out4) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) reinterpret_cast<int4*>(out4)[ids[i]] = reinterpret_cast<const int4*>(rows)[i];
}
hipError_t launch_scatter_rows(const int32_t* rows, const uint32_t* ids, uint32_t m, int32_t* out4, hipStream_t st) {
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_rows, dim3((m + 255) / 256), dim3(256), 0, st, rows, ids, m, out4);
    return hipGetLastError();
}
hipError_t launch_fill_reroute(int32_t* out4, uint32_t nreads, hipStream_t st) {
    if (nreads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_reroute, dim3((nreads + 255) / 256), dim3(256), 0, st, out4, nreads);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_max_len(const uint32_t* __restrict__ offs, uint32_t nreads, uint32_t* d_max) {
    uint32_t m = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nreads; i += gridDim.x * 256) {
        uint32_t l = offs[i + 1] - offs[i];
        m = l > m ? l : m;
    }
    m = (uint32_t)wave_max_i32((int)m);
    // one atomic per block, and only from blocks that would raise the value: atomics on ONE address are served one at a time
    // (8192 of them took 96 us for 1 M reads)
    __shared__ uint32_t wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = wm[i] > m ? wm[i] : m;
        if (m > __hip_atomic_load(d_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d_max, m);
    }
}
hipError_t launch_max_len(const uint32_t* offs, uint32_t nreads, uint32_t* d_max, hipStream_t st) {
    hipError_t e = hipMemsetAsync(d_max, 0, 4, st);
    if (e != hipSuccess) return e;
    if (nreads == 0) return hipSuccess;
    uint32_t grid = (nreads + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_max_len, dim3(grid), dim3(256), 0, st, offs, nreads, d_max);
    return hipGetLastError();
}

// mask_by_frequency (rkmh.cpp:916): order-preserving, in place
__global__ __launch_bounds__(256) void k_mask_by_frequency(uint64_t* h, uint64_t n, const int32_t* __restrict__ counter,
                                                           uint64_t slots, int min_occ, int strict) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = counter[h[i] % slots];
    if (strict ? (c < min_occ) : (c <= min_occ)) h[i] = 0;
}
hipError_t launch_mask_by_frequency(uint64_t* h, uint64_t n, const int32_t* counter, uint64_t slots, int min_occ,
                                    const DevPolicy& pol, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mask_by_frequency, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, h, n, counter, slots, min_occ,
                       pol.mask_strict_less);
    return hipGetLastError();
}

// set<hash_t> sample_set(hashes...) ; for (x : sample_set) counter.increment(x)   (rkmh.cpp:348-355)
// key 0 (the invalid-k-mer sentinel, a member of the set like any other value) is tracked by a flag word.
__global__ __launch_bounds__(256) void k_count_distinct(const uint64_t* __restrict__ h, uint64_t n, unsigned long long* table,
                                                        uint64_t tmask, int32_t* counter, uint64_t slots, unsigned int* zero_seen) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = h[i];
        if (v == 0) {
            if (atomicExch(zero_seen, 1u) == 0u) atomicAdd(&counter[0], 1);
            continue;
        }
        uint64_t s = (v * 0x9E3779B97F4A7C15ull) >> 20 & tmask;
        for (;;) {
            const unsigned long long old = atomicCAS(&table[s], 0ull, (unsigned long long)v);
            if (old == 0ull) { atomicAdd(&counter[v % slots], 1); break; }
            if (old == v) break;
            s = (s + 1) & tmask;
        }
    }
}
hipError_t launch_count_distinct(const uint64_t* hashes, uint64_t n, uint64_t* table, uint64_t tsize, int32_t* counter,
                                 uint64_t slots, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(table, 0, (tsize + 1) * 8, st); // last word = the zero-seen flag
    if (e != hipSuccess) return e;
    uint32_t grid = (uint32_t)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_count_distinct, dim3(grid), dim3(256), 0, st, hashes, n, reinterpret_cast<unsigned long long*>(table),
                       tsize - 1, counter, slots, reinterpret_cast<unsigned int*>(table + tsize));
    return hipGetLastError();
}

__global__ void k_counter_inc(int32_t* counter, uint64_t slots, uint64_t key) { atomicAdd(&counter[key % slots], 1); }
// dst[i] += src[i]: the sum of two devices' (or two passes') depth tables -- the all-reduce step of a multi-device -M run
// ALIGNED: both tables start on a 16-byte boundary (every table the library allocates does; a table adopted with rk_counter_wrap --
// a slice of a torch tensor, say -- may be only 4-byte aligned and takes the dword form)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_counter_add(int32_t* __restrict__ dst, const int32_t* __restrict__ src, uint64_t n) {
    if (!ALIGNED) {
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] += src[i];
        return;
    }
    const uint64_t n4 = n >> 2;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
        int4 a = reinterpret_cast<int4*>(dst)[i];
        const int4 b = reinterpret_cast<const int4*>(src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<int4*>(dst)[i] = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] += src[(n4 << 2) + threadIdx.x];
}
hipError_t launch_counter_add(int32_t* dst, const int32_t* src, uint64_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) != 0) {
        uint64_t b = (n + 255) / 256;
        if (b > 256 * 32) b = 256 * 32;
        hipLaunchKernelGGL(k_counter_add<false>, dim3((uint32_t)b), dim3(256), 0, st, dst, src, n);
        return hipGetLastError();
    }
    uint64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_counter_add<true>, dim3((uint32_t)blocks), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}
hipError_t launch_counter_inc(int32_t* counter, uint64_t slots, uint64_t key, hipStream_t st) {
    hipLaunchKernelGGL(k_counter_inc, dim3(1), dim3(1), 0, st, counter, slots, key);
    return hipGetLastError();
}

} // namespace rk
