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
        printf("%s rc=%d sum=%llu\n", argv[i], rc, sum);
    }
    return 0;
}
