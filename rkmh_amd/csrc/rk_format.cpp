// rk_format.cpp -- the output of stream / classify / filter for one block of the device FASTQ front end (rk_fastq_slot_*), written
// from the record names (sequences, quality strings) WHERE THEY LIE in the block's raw text: the host side of the per-read loop of
// /root/reference/src/rkmh.cpp:845-898 shrinks to this.  Line format: rkmh.cpp:887-892; filter's records: rkmh.cpp:1292-1300 with
// the decision of classify_and_count_diff_filter (/root/reference/src/equiv.hpp:324-353).  Used by bin/rkmh (rkmh_main.cpp) and,
// through ctypes, by the one-process-per-GPU front end (rkmh_amd/cli.py).
#include "../../include/rkmh_amd.h"
#include "rk_filter_rule.hpp"

#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

extern "C" void rk__set_error(const char* msg); // rk_api.hip

// Everything of a line that does not depend on the read, prepared once per run: "ref name \t" per reference and the eight
// possible tails "<sketch>[FAIL:DEPTH] \t [FAIL:MATCHES] \t [FAIL:DIFF] \n".
struct rk_line_parts {
    std::vector<char> ref_text;          // padded: copies run in 16-byte steps
    std::vector<uint32_t> ref_off, ref_len;
    char tail[8][48];
    uint32_t tail_len[8];
    size_t maxref = 0;
    int min_matches = -1, min_diff = 0;
};

namespace {

inline char* put_int(char* w, int v) {
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *w++ = '-';
    while (n) *w++ = tmp[--n];
    return w;
}
// copies n bytes in 16-byte steps (both buffers have the slack): no call into memcpy for a ten-byte name
inline char* copy16(char* w, const char* src, size_t n) {
    for (size_t i = 0; i < n; i += 16) memcpy(w + i, src + i, 16);
    return w + n;
}
int bad(const char* m) { rk__set_error(m); return RK_ERR_ARG; }

} // namespace

extern "C" int rk_line_parts_create(const char* ref_names, const uint64_t* name_offsets, int64_t nref, int sketch_size, int min_matches,
                                    int min_diff, rk_line_parts** out) {
    if (!ref_names || !name_offsets || nref < 1 || !out) return bad("rk_line_parts_create: bad arguments");
    rk_line_parts* lp = new (std::nothrow) rk_line_parts();
    if (!lp) return RK_ERR_NOMEM;
    lp->min_matches = min_matches; lp->min_diff = min_diff;
    lp->ref_off.resize((size_t)nref); lp->ref_len.resize((size_t)nref);
    for (int64_t r = 0; r < nref; ++r) {
        const size_t ln = (size_t)(name_offsets[r + 1] - name_offsets[r]) - 1; // offsets include the NUL
        lp->ref_off[(size_t)r] = (uint32_t)lp->ref_text.size(); lp->ref_len[(size_t)r] = (uint32_t)ln + 1;
        lp->ref_text.insert(lp->ref_text.end(), ref_names + name_offsets[r], ref_names + name_offsets[r] + ln);
        lp->ref_text.push_back('\t');
        if (ln + 1 > lp->maxref) lp->maxref = ln + 1;
    }
    lp->ref_text.resize(lp->ref_text.size() + 32, 0);
    for (int f = 0; f < 8; ++f) {
        char* w = put_int(lp->tail[f], sketch_size);
        if (f & 1) { memcpy(w, "FAIL:DEPTH", 10); w += 10; }
        *w++ = '\t';
        if (f & 2) { memcpy(w, "FAIL:MATCHES", 12); w += 12; }
        *w++ = '\t';
        if (f & 4) { memcpy(w, "FAIL:DIFF", 9); w += 9; }
        *w++ = '\n';
        lp->tail_len[f] = (uint32_t)(w - lp->tail[f]);
    }
    *out = lp;
    return RK_OK;
}
extern "C" void rk_line_parts_destroy(rk_line_parts* lp) { delete lp; }

extern "C" uint64_t rk_fastq_stream_lines_bound(const rk_line_parts* lp, const rk_fastq_result* r) {
    if (!lp || !r || r->status != 0) return 64;
    uint64_t names = 0;
    for (int64_t i = 0; i < r->nrec; ++i) names += r->name_len[i];
    return names + (uint64_t)r->nrec * (lp->maxref + 64) + 64;
}

// the lines of one block; text = the block as it was given to the slot (rk_fastq_slot_text: readable 16 bytes past every name)
extern "C" int64_t rk_fastq_stream_lines(const rk_line_parts* lp, const rk_fastq_result* r, const uint8_t* text, char* dst, uint64_t cap) {
    if (!lp || !r || !text || !dst) return bad("rk_fastq_stream_lines: bad arguments");
    if (r->status != 0) return bad("rk_fastq_stream_lines: the block was refused by the device (status != 0)");
    if (cap < rk_fastq_stream_lines_bound(lp, r)) return bad("rk_fastq_stream_lines: buffer smaller than rk_fastq_stream_lines_bound");
    const size_t nref = lp->ref_off.size();
    char* w = dst;
    for (int64_t i = 0; i < r->nrec; ++i) {
        const int32_t* q = r->out4 + i * 4;
        if ((uint32_t)q[0] >= nref) return bad("rk_fastq_stream_lines: reference index outside the panel");
        w = copy16(w, lp->ref_text.data() + lp->ref_off[(size_t)q[0]], lp->ref_len[(size_t)q[0]]);
        w = copy16(w, (const char*)text + r->name_off[i], r->name_len[i]); *w++ = '\t';
        w = put_int(w, q[1]); *w++ = '\t';
        const int f = (q[3] <= lp->min_matches ? 1 : 0) | (q[1] < lp->min_matches ? 2 : 0) | (!(q[2] > lp->min_diff) ? 4 : 0);
        memcpy(w, lp->tail[f], 48);
        w += lp->tail_len[f];
    }
    return (int64_t)(w - dst);
}

extern "C" uint64_t rk_fastq_filter_records_bound(const rk_fastq_result* r) {
    if (!r || r->status != 0) return 64;
    uint64_t need = 64;
    for (int64_t i = 0; i < r->nrec; ++i) need += (uint64_t)r->name_len[i] + 2 * (uint64_t)r->seq_len[i] + 8;
    return need;
}

// filter's output for one block: ">name \n SEQ \n + \n QUAL \n" of every read that passes (SEQ upper-cased as parse_fastas does, rkmh.cpp:280)
extern "C" int64_t rk_fastq_filter_records(const rk_fastq_result* r, const uint8_t* text, int min_matches, int min_diff, char* dst, uint64_t cap) {
    if (!r || !text || !dst) return bad("rk_fastq_filter_records: bad arguments");
    if (r->status != 0) return bad("rk_fastq_filter_records: the block was refused by the device (status != 0)");
    if (cap < rk_fastq_filter_records_bound(r)) return bad("rk_fastq_filter_records: buffer smaller than rk_fastq_filter_records_bound");
    char* w = dst;
    for (int64_t i = 0; i < r->nrec; ++i) {
        if (!rk_filter_keeps(r->out4 + i * 4, min_matches, min_diff)) continue; // (rk_filter_rule.hpp: equiv.hpp:324-353, rkmh.cpp:1292-1293)
        *w++ = '>';
        memcpy(w, text + r->name_off[i], r->name_len[i]); w += r->name_len[i];
        *w++ = '\n';
        const uint8_t* sq = text + r->seq_off[i];
        const uint32_t n = r->seq_len[i];
        for (uint32_t j = 0; j < n; ++j) { const signed char ch = (signed char)sq[j]; w[j] = (char)(((int)ch - 91) > 0 ? ch - 32 : ch); }
        w += n;
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        memcpy(w, text + r->qual_off[i], n); w += n;
        *w++ = '\n';
    }
    return (int64_t)(w - dst);
}
