#include "../../include/rkmh_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
extern "C" void rk__set_error(const char* m) { (void)m; }
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        const char* paths[1] = {argv[i]};
        rk_seqset s;
        int rc = rk_parse_files(paths, 1, &s);
        unsigned long long sum = 0;
        if (rc == 0) {
            for (long long j = 0; j < s.nseq; ++j) sum += s.offsets[j + 1] - s.offsets[j] + strlen(s.names + s.name_offsets[j]);
            rk_seqset_free(&s);
        }
        rk_reader* r = nullptr;
        if (rk_reader_open(argv[i], &r) == 0) {
            for (;;) { rk_seqset b; if (rk_reader_next(r, 1 << 16, 1 << 20, &b) != 0) break; if (b.nseq == 0) { rk_seqset_free(&b); break; } sum += b.nseq; rk_seqset_free(&b); }
            rk_reader_close(r);
        }
        if (rk_reader_open(argv[i], &r) == 0) {
            for (;;) { rk_seqset b; if (rk_reader_next(r, 7, 0, &b) != 0) break; if (b.nseq == 0) { rk_seqset_free(&b); break; } sum += b.nseq; rk_seqset_free(&b); }
            rk_reader_close(r);
        }
        // the BGZF member-table parser and record cutter on the same bytes (gen.py writes *.bgzf.txt images, some of them damaged):
        // rk_bgzf_open refuses what is not BGZF; what it accepts is planned, inflated and cut job by job
        rk_bgzf* z = nullptr;
        if (rk_bgzf_open(argv[i], &z) == 0) {
            const int64_t nm = rk_bgzf_members(z);
            int64_t first[64];
            const int64_t nj = rk_bgzf_plan(z, 1 << 16, first, 64);
            unsigned char* dst = (unsigned char*)malloc(((size_t)1 << 22) + 64);
            sum += (unsigned long long)rk_bgzf_first_byte(z) + (unsigned long long)rk_bgzf_lead_member(z, nm / 2);
            for (int64_t j = 0; j < nj; ++j) {
                uint64_t nb = 0, off = 0;
                const int rcz = rk_bgzf_fastq_records(z, first[j], first[j + 1], dst, (size_t)1 << 22, &nb, &off);
                sum += (unsigned long long)(rcz == 0 ? nb + off : (unsigned long long)(100 + rcz));
            }
            free(dst);
            rk_bgzf_close(z);
        }
        printf("%s rc=%d sum=%llu\n", argv[i], rc, sum);
    }
    return 0;
}
