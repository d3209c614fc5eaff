// rk_parse.cpp -- FASTA/FASTQ(.gz) front end of the hot path (host side).
//
// Replaces parse_fastas (/root/reference/src/rkmh.cpp:238-292), which drives Heng Li's kseq macros
// through gzFile.  The record grammar below follows kseq_read (/root/reference/src/kseq.hpp:170-208)
// decision for decision -- records start at the next '>' or '@' wherever it is, the name ends at the
// first whitespace, sequence bytes are every isgraph() byte up to the next '>', '+' or '@', a FASTQ
// quality string is read by COUNT (so '@' inside it is data) and a short one ends the file (-2 ends the
// caller's loop, rkmh.cpp:251) -- but it is a block scanner over a large zlib buffer that appends straight
// into one concatenated batch (bases + offsets) ready for hipMemcpyAsync, not a per-record kstring.
// Bases are NOT upper-cased here: the device does that while staging (rkmh.cpp:252 / :856).
#include "../../include/rkmh_amd.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {


struct Tables {
    unsigned char seqcls[256]; // 0 skip, 1 keep, 2 terminator
    unsigned char space[256];
    Tables() {
        for (int c = 0; c < 256; ++c) {
            seqcls[c] = (c >= 33 && c <= 126) ? 1 : 0;
            space[c] = (c == ' ' || (c >= 9 && c <= 13)) ? 1 : 0;
        }
        seqcls['>'] = seqcls['+'] = seqcls['@'] = 2;
    }
};
const Tables T;

template <typename V> struct Grow { // malloc-backed growable array handed over to C callers
    V* p = nullptr;
    size_t n = 0, cap = 0;
    bool reserve(size_t want) {
        if (want <= cap) return true;
        size_t nc = cap ? cap : 1024;
        while (nc < want) nc += nc >> 1;
        V* q = (V*)realloc(p, nc * sizeof(V));
        if (!q) return false;
        p = q; cap = nc;
        return true;
    }
    bool push(V v) { if (n == cap && !reserve(n + 1)) return false; p[n++] = v; return true; }
    bool append(const V* s, size_t k) { if (!reserve(n + k)) return false; memcpy(p + n, s, k * sizeof(V)); n += k; return true; }
    V* release() { V* r = p; p = nullptr; n = cap = 0; return r; }
    ~Grow() { free(p); }
};

struct Batch {
    Grow<uint8_t> bases;
    Grow<uint64_t> offsets;
    Grow<char> names;
    Grow<uint64_t> name_offsets;
    Grow<char> quals;
    bool quals_ok = true; // every record so far had a quality string
    int64_t nseq = 0;
    bool init() { return offsets.push(0) && name_offsets.push(0); }
};

} // namespace

struct rk_reader {
    gzFile fp = nullptr;
    std::vector<unsigned char> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    int last_char = 0;
    bool finished = false;

    inline bool fill() {
        if (eof) return false;
        int r = gzread(fp, buf.data(), (unsigned)buf.size());
        beg = 0;
        end = r > 0 ? (size_t)r : 0;
        if (end < buf.size()) eof = true;
        return end > 0;
    }
    inline int getc() {
        if (beg >= end && !fill()) return -1;
        return buf[beg++];
    }

    // one record appended to b. returns 1 record read, 0 clean EOF, -2 truncated, -3 out of memory
    int next(Batch& b) {
        int c;
        if (last_char == 0) {
            for (;;) { // jump to the next header char
                if (beg >= end && !fill()) return 0;
                const unsigned char* p = buf.data() + beg;
                const unsigned char* e = buf.data() + end;
                while (p < e && *p != '>' && *p != '@') ++p;
                beg = (size_t)(p - buf.data());
                if (p < e) { last_char = *p; ++beg; break; }
            }
        }
        // name = up to the first whitespace
        const size_t name0 = b.names.n;
        bool any = false;
        c = -1;
        for (;;) {
            if (beg >= end && !fill()) { c = -1; break; }
            any = true;
            const unsigned char* p = buf.data() + beg;
            const unsigned char* e = buf.data() + end;
            const unsigned char* q = p;
            while (q < e && !T.space[*q]) ++q;
            if (!b.names.append((const char*)p, (size_t)(q - p))) return -3;
            beg = (size_t)(q - buf.data());
            if (q < e) { c = *q; ++beg; break; }
        }
        if (!any) { b.names.n = name0; return 0; } // EOF right after a header char (ks_getuntil < 0)
        if (c != '\n' && c != -1) {               // comment: rest of the header line
            for (;;) {
                if (beg >= end && !fill()) break;
                const unsigned char* p = buf.data() + beg;
                const void* nl = memchr(p, '\n', end - beg);
                if (nl) { beg = (size_t)((const unsigned char*)nl - buf.data()) + 1; break; }
                beg = end;
            }
        }
        // sequence
        const size_t seq0 = b.bases.n;
        c = -1;
        for (;;) {
            if (beg >= end && !fill()) { c = -1; break; }
            const unsigned char* p = buf.data() + beg;
            const unsigned char* e = buf.data() + end;
            if (!b.bases.reserve(b.bases.n + (size_t)(e - p))) return -3;
            uint8_t* w = b.bases.p + b.bases.n;
            unsigned char cls = 0;
            while (p < e) {
                // fast path: copy a line's worth of keepers
                cls = T.seqcls[*p];
                if (cls == 1) { *w++ = *p++; continue; }
                if (cls == 2) break;
                ++p;
            }
            b.bases.n = (size_t)(w - b.bases.p);
            beg = (size_t)(p - buf.data());
            if (p < e) { c = *p; ++beg; break; }
        }
        const size_t slen = b.bases.n - seq0;
        if (c == '>' || c == '@') last_char = c;
        bool fastq = (c == '+');
        if (fastq) {
            for (;;) { // rest of the '+' line
                c = getc();
                if (c == -1 || c == '\n') break;
            }
            if (c == -1) { b.bases.n = seq0; b.names.n = name0; return -2; }
            const size_t q0 = b.quals.n;
            size_t ql = 0;
            if (b.quals_ok && !b.quals.reserve(q0 + slen + 1)) return -3;
            for (;;) {
                c = getc();
                if (c == -1 || !(ql < slen)) break;
                if (c >= 33 && c <= 127) { if (b.quals_ok) b.quals.p[q0 + ql] = (char)c; ++ql; }
            }
            last_char = 0;
            if (ql != slen) { b.bases.n = seq0; b.names.n = name0; return -2; }
            if (b.quals_ok) b.quals.n = q0 + slen;
        } else {
            b.quals_ok = false;
        }
        if (!b.names.push('\0')) return -3;
        if (!b.offsets.push((uint64_t)b.bases.n) || !b.name_offsets.push((uint64_t)b.names.n)) return -3;
        ++b.nseq;
        return 1;
    }
};

extern "C" void rk__set_error(const char* msg); // rk_api.hip
static int perr(int code, const std::string& m) { rk__set_error(m.c_str()); return code; }

static int hand_over(Batch& b, rk_seqset* out) {
    if (!b.bases.reserve(b.bases.n + 64)) return perr(RK_ERR_NOMEM, "out of memory");
    memset(b.bases.p + b.bases.n, 0, 64); // dword-read slack for the device staging loads
    out->nseq = b.nseq;
    bool have_q = b.quals_ok && b.nseq > 0;
    out->bases = b.bases.release();
    out->offsets = b.offsets.release();
    out->names = b.names.release();
    out->name_offsets = b.name_offsets.release();
    out->quals = have_q ? b.quals.release() : nullptr;
    return RK_OK;
}

extern "C" {

int rk_reader_open(const char* path, rk_reader** out) {
    if (!path || !out) return perr(RK_ERR_ARG, "bad arguments");
    gzFile fp = strcmp(path, "-") == 0 ? gzdopen(0, "r") : gzopen(path, "r");
    if (!fp) return perr(RK_ERR_IO, std::string("cannot open ") + path);
    gzbuffer(fp, 1 << 20);
    rk_reader* r = new rk_reader();
    r->fp = fp;
    r->buf.resize(4u << 20);
    *out = r;
    return RK_OK;
}

void rk_reader_close(rk_reader* r) {
    if (!r) return;
    if (r->fp) gzclose(r->fp);
    delete r;
}

// Reads up to max_records records / max_bases bases (0 = unlimited) into *out (malloc'd arrays; release with
// rk_seqset_free).  out->nseq == 0 means end of input.
int rk_reader_next(rk_reader* r, int64_t max_records, uint64_t max_bases, rk_seqset* out) {
    if (!r || !out) return perr(RK_ERR_ARG, "bad arguments");
    memset(out, 0, sizeof *out);
    Batch b;
    if (!b.init()) return perr(RK_ERR_NOMEM, "out of memory");
    while (!r->finished) {
        if (max_records > 0 && b.nseq >= max_records) break;
        if (max_bases > 0 && b.bases.n >= max_bases) break;
        int rc = r->next(b);
        if (rc == -3) return perr(RK_ERR_NOMEM, "out of memory");
        if (rc <= 0) { r->finished = true; break; } // EOF, or truncated record: ends the file (rkmh.cpp:251)
    }
    return hand_over(b, out);
}

int rk_parse_files(const char* const* paths, int npaths, rk_seqset* out) {
    if (!paths || npaths < 0 || !out) return perr(RK_ERR_ARG, "bad arguments");
    memset(out, 0, sizeof *out);
    Batch b;
    if (!b.init()) return perr(RK_ERR_NOMEM, "out of memory");
    for (int i = 0; i < npaths; ++i) {
        rk_reader* r = nullptr;
        int rc = rk_reader_open(paths[i], &r);
        if (rc != RK_OK) return rc;
        for (;;) {
            int k = r->next(b);
            if (k == -3) { rk_reader_close(r); return perr(RK_ERR_NOMEM, "out of memory"); }
            if (k <= 0) break;
        }
        rk_reader_close(r);
    }
    return hand_over(b, out);
}

void rk_seqset_free(rk_seqset* s) {
    if (!s) return;
    free(s->bases); free(s->offsets); free(s->names); free(s->name_offsets); free(s->quals);
    memset(s, 0, sizeof *s);
}

} // extern "C"
