// rk_synth.cpp -- host-side generator of the synthetic read workload.
// Same stream as rkmh_amd/synth.py (which documents the draw plan and is the tested definition); this C++
// twin exists because bench.py needs 10^6..10^7 reads per rank in seconds, not minutes.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rkmh_amd.h"

namespace {
inline uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t draw(uint64_t st, uint64_t j) { return mix(st + 0x9E3779B97F4A7C15ULL * (j + 1)); }
inline uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
inline int code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
inline uint8_t comp(uint8_t c) { return c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c; }
} // namespace

extern "C" int rk_synth_reads(const uint8_t* ref_bases, const uint64_t* ref_offsets, int nref, uint64_t lo, uint64_t hi,
                              int read_len, uint64_t seed, uint8_t* out /* [(hi-lo)*read_len] */, int threads) {
    if (!ref_bases || !ref_offsets || nref < 1 || hi < lo || read_len < 1 || !out) return RK_ERR_ARG;
    for (int r = 0; r < nref; ++r)
        if (ref_offsets[r + 1] - ref_offsets[r] < (uint64_t)read_len) return RK_ERR_ARG;
    if (threads < 1) threads = 1;
    const uint64_t n = hi - lo;
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    auto work = [&](uint64_t a, uint64_t b) {
        std::vector<uint8_t> tmp((size_t)read_len);
        for (uint64_t i = a; i < b; ++i) {
            const uint64_t st = seed + (lo + i);
            const uint64_t ref = draw(st, 0) % (uint64_t)nref;
            const uint64_t len = ref_offsets[ref + 1] - ref_offsets[ref];
            const uint64_t start = draw(st, 1) % (len - (uint64_t)read_len + 1) + ref_offsets[ref];
            const bool flip = draw(st, 2) & 1;
            const bool hasn = draw(st, 3) % 1000 == 0;
            const uint64_t npos = draw(st, 4) % (uint64_t)read_len;
            uint8_t* dst = out + i * (uint64_t)read_len;
            for (int j = 0; j < read_len; ++j) {
                uint8_t c = up(ref_bases[start + (uint64_t)j]);
                const uint64_t d = draw(st, 5 + (uint64_t)j);
                if (d % 100 == 0) {
                    int cd = code(c);
                    if (cd < 4) c = (uint8_t)acgt[((uint64_t)cd + 1 + ((d >> 32) % 3)) % 4];
                }
                tmp[(size_t)j] = c;
            }
            if (flip) for (int j = 0; j < read_len; ++j) dst[j] = comp(tmp[(size_t)(read_len - 1 - j)]);
            else memcpy(dst, tmp.data(), (size_t)read_len);
            if (hasn) dst[npos] = 'N';
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) {
        uint64_t a = n * (uint64_t)t / (uint64_t)threads, b = n * (uint64_t)(t + 1) / (uint64_t)threads;
        th.emplace_back(work, a, b);
    }
    for (auto& x : th) x.join();
    return RK_OK;
}
