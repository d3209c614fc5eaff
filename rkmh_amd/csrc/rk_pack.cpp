// rk_pack.cpp -- packed reads on the host: 2 bits per base + an exception list (include/rkmh_amd.h, "PACKED READS").  What `rkmh pack`
// writes and what filter -F decodes for the reads it prints.  The reference parses -F/--pre-reads and stops there
// (/root/reference/src/rkmh.cpp:659-664); mkmh's to_upper (src/rkmh.cpp:856) folds a..z onto A..Z before anything hashes, so the case
// of acgt carries no information on this path and is not kept.
#include "../../include/rkmh_amd.h"

#include <cstring>

extern "C" void rk__set_error(const char* msg);

extern "C" int64_t rk_packed_encode(const uint8_t* bases, uint64_t n, uint64_t base_index0, uint8_t* bases2, rk_packed_exception* exc, uint64_t cap) {
    if ((!bases && n) || !bases2) { rk__set_error("rk_packed_encode: bad arguments"); return RK_ERR_ARG; }
    uint64_t ne = 0;
    memset(bases2, 0, (size_t)((n + 3) / 4));
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t c = bases[i], u = c & 0xDFu; // (letters: upper case)
        uint32_t code = (c >> 1) & 3u;               // A 0, C 1, T 2, G 3
        if (u != 'A' && u != 'C' && u != 'G' && u != 'T') {
            if (ne >= cap || !exc) { rk__set_error("rk_packed_encode: more exceptions than the caller's array holds"); return RK_ERR_LIMIT; }
            exc[ne].pos = (uint32_t)(base_index0 + i); exc[ne].byte = c;
            ++ne;
            code = 0;
        }
        bases2[i >> 2] |= (uint8_t)(code << (2u * (uint32_t)(i & 3u)));
    }
    return (int64_t)ne;
}

extern "C" void rk_packed_decode(const uint8_t* bases2, uint64_t first, uint64_t n, const rk_packed_exception* exc, uint32_t nexc, uint8_t* out) {
    static const char L[4] = {'A', 'C', 'T', 'G'};
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t p = first + i;
        out[i] = (uint8_t)L[(bases2[p >> 2] >> (2u * (uint32_t)(p & 3u))) & 3u];
    }
    // exceptions in [first, first + n): binary search for the first one
    uint32_t lo = 0, hi = nexc;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (exc[mid].pos < first) lo = mid + 1; else hi = mid; }
    for (; lo < nexc && exc[lo].pos < first + n; ++lo) out[exc[lo].pos - first] = (uint8_t)exc[lo].byte;
}

#include "rk_filter_rule.hpp"

// filter's output (src/rkmh.cpp:1292-1300) for one classified block of a packed file: ">name \n SEQ \n + \n QUAL \n" of every read that
// passes (rk_filter_keeps); SEQ decoded from the 2-bit bases and upper-cased as parse_fastas does (rkmh.cpp:280); QUAL from the file, or
// empty when it keeps none (as for reads that came from FASTA).
extern "C" uint64_t rk_packed_filter_records_bound(const rk_fastq_result* r) {
    if (!r) return 64;
    uint64_t need = 64;
    for (int64_t i = 0; i < r->nrec; ++i) need += (uint64_t)r->name_len[i] + 2 * (uint64_t)r->seq_len[i] + 8;
    return need;
}
extern "C" int64_t rk_packed_filter_records(const rk_fastq_result* r, const rk_packed_block* b, const uint8_t* file, int min_matches, int min_diff, char* dst, uint64_t cap) {
    if (!r || !b || !file || !dst) { rk__set_error("rk_packed_filter_records: bad arguments"); return RK_ERR_ARG; }
    if (cap < rk_packed_filter_records_bound(r)) { rk__set_error("rk_packed_filter_records: buffer smaller than rk_packed_filter_records_bound"); return RK_ERR_ARG; }
    const uint8_t* names = file + b->names_off;
    const uint8_t* bases2 = file + b->bases_off;
    const rk_packed_exception* exc = reinterpret_cast<const rk_packed_exception*>(file + b->exc_off);
    const uint8_t* quals = b->quals_off ? file + b->quals_off : nullptr;
    char* w = dst;
    for (int64_t i = 0; i < r->nrec; ++i) {
        if (!rk_filter_keeps(r->out4 + i * 4, min_matches, min_diff)) continue;
        *w++ = '>';
        memcpy(w, names + r->name_off[i], r->name_len[i]); w += r->name_len[i];
        *w++ = '\n';
        const uint32_t n = r->seq_len[i];
        rk_packed_decode(bases2, r->seq_off[i], n, exc, b->nexc, reinterpret_cast<uint8_t*>(w));
        for (uint32_t j = 0; j < n; ++j) { const signed char ch = (signed char)w[j]; w[j] = (char)(((int)ch - 91) > 0 ? ch - 32 : ch); }
        w += n;
        *w++ = '\n'; *w++ = '+'; *w++ = '\n';
        if (quals) { memcpy(w, quals + r->seq_off[i], n); w += n; }
        *w++ = '\n';
    }
    return (int64_t)(w - dst);
}
