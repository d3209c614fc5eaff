// Pins rk_crc32.hpp (the tables and the way a wave's lanes share a member's text) against zlib's crc32 on the host:
//   g++ -O2 -std=c++17 -Irkmh_amd/csrc tools/crc32_check.cpp -lz -o /tmp/crc32_check && /tmp/crc32_check
// The device kernel (k_crc32_members, rk_inflate.hip) runs the same crc32_piece / crc32_advance per lane.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rk_crc32.hpp"

int main() {
    static constexpr rk::Crc32Tables T = rk::make_crc32_tables();
    const uint32_t* zt = (const uint32_t*)get_crc_table();
    for (int i = 0; i < 256; ++i)
        if (T.byte[i] != zt[i]) { printf("byte table differs at %d\n", i); return 1; }
    std::vector<unsigned char> buf(1 << 20);
    srand(7);
    for (auto& b : buf) b = (unsigned char)(rand() >> 7);
    const uint32_t lens[] = {0, 1, 2, 15, 16, 17, 1023, 1024, 1039, 1040, 1041, 2080, 5000, 40000, 65279, 65280, 65535, 65536};
    long checked = 0;
    for (uint32_t n : lens)
        for (uint32_t a : {0u, 1u, 7u, 15u, 16u, 17u, 1039u, 4097u, 70001u, 500000u}) {
            uint32_t total = 0, covered = 0;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const rk::Crc32Piece p = rk::crc32_piece(a, n, lane);
                uint32_t s = p.init;
                for (uint32_t q = p.b; q < p.e; ++q) s = T.byte[(s ^ buf[q]) & 0xFFu] ^ (s >> 8);
                covered += p.e - p.b;
                if (p.e > p.b || lane == 0) total ^= rk::crc32_advance(T, s, p.z);
                if (lane > 0 && p.e > p.b && (p.b & 15u)) { printf("piece not aligned: a %u n %u lane %u\n", a, n, lane); return 1; }
            }
            const uint32_t want = (uint32_t)crc32(0L, buf.data() + a, n);
            if (covered != n || ~total != want) { printf("MISMATCH a %u n %u: covered %u, got %08x want %08x\n", a, n, covered, ~total, want); return 1; }
            ++checked;
        }
    printf("crc32 pieces: %ld (offset, length) pairs agree with zlib\n", checked);
    return 0;
}
