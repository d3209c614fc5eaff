// rk_crc32.hpp -- CRC-32 (the gzip polynomial, reflected 0xEDB88320) split over the lanes of a wave.
//
// gzread verifies every member's CRC-32 (the reference reads all input through it, /root/reference/src/rkmh.cpp:238-263); the device
// inflater (rk_inflate.hip) does the same for the text it builds.  A CRC register is a linear function of (register, message) over
// GF(2): for a message cut into pieces P_0 .. P_m,
//     state(init, P_0 .. P_m) = XOR_i  advance(x_i, bytes behind P_i),   x_0 = state(init, P_0),  x_i = state(0, P_i)  (i > 0),
// where advance(s, z) runs z zero bytes through the register -- a 32 x 32 bit matrix A^z.  The tables below hold the byte step
// (`byte`: the classic 256-entry table) and A^(2^k) for k < 17 (`adv[k][b]` = the image of bit b), so a lane advances its piece's
// register past up to 128 KB of text in at most 17 matrix products.  constexpr: the tables are compile-time constants on both sides
// (the host uses them in tools/crc32_check.cpp to pin them against zlib).
#pragma once
#include <cstdint>

namespace rk {

struct Crc32Tables {
    uint32_t byte[256];
    uint32_t adv[17][32];
};

constexpr uint32_t crc32_zero_byte_step(const uint32_t (&byte)[256], uint32_t s) { return byte[s & 0xFFu] ^ (s >> 8); }

constexpr Crc32Tables make_crc32_tables() {
    Crc32Tables t{};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int j = 0; j < 8; ++j) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        t.byte[i] = c;
    }
    for (int b = 0; b < 32; ++b) t.adv[0][b] = crc32_zero_byte_step(t.byte, 1u << b); // one zero byte
    for (int k = 1; k < 17; ++k)
        for (int b = 0; b < 32; ++b) { // A^(2^k) = A^(2^(k-1)) applied twice
            const uint32_t v = t.adv[k - 1][b];
            uint32_t r = 0;
            for (int i = 0; i < 32; ++i)
                if ((v >> i) & 1u) r ^= t.adv[k - 1][i];
            t.adv[k][b] = r;
        }
    return t;
}

#if defined(__HIPCC__)
#define RK_CRC_HD __host__ __device__
#else
#define RK_CRC_HD
#endif

// How the 64 lanes of a wave share the text [a, a + n) of one member (n <= 65536; a = its offset in the text buffer): pieces of
// 1040 bytes counted from the 16-byte boundary at or below a, so every piece but the first begins on one (the lanes read whole
// aligned 16-byte groups).  Lane 0's piece begins at a and starts from the CRC's initial register; lanes past the end hold nothing.
constexpr uint32_t CRC_PIECE = 1040; // 64 x 1040 >= 65536 + 15
struct Crc32Piece { uint32_t b, e, init, z; }; // bytes [b, e); register before the piece; zero bytes behind it
RK_CRC_HD inline Crc32Piece crc32_piece(uint32_t a, uint32_t n, uint32_t lane) {
    const uint32_t a0 = a & ~15u, end = a + n, lo = a0 + CRC_PIECE * lane, hi = lo + CRC_PIECE;
    Crc32Piece p;
    p.b = lo < a ? a : lo;
    p.e = hi < end ? hi : end;
    if (p.b > end) p.b = end;
    if (p.e < p.b) p.e = p.b;
    p.init = lane == 0u ? 0xFFFFFFFFu : 0u;
    p.z = end - p.e;
    return p;
}

// the register after z zero bytes
RK_CRC_HD inline uint32_t crc32_advance(const Crc32Tables& t, uint32_t s, uint32_t z) {
    for (int k = 0; k < 17 && (z >> k) != 0u; ++k) {
        if (!((z >> k) & 1u)) continue;
        uint32_t r = 0;
        for (int b = 0; b < 32; ++b) r ^= (0u - ((s >> b) & 1u)) & t.adv[k][b];
        s = r;
    }
    return s;
}

} // namespace rk
