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

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <memory>
#include <mutex>
#include <unordered_map>
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

// Recycling of the big batch buffers.  A 256 MB batch handed to the caller comes back through rk_seqset_free a few
// milliseconds later; glibc serves and releases blocks of that size with mmap / munmap, so every batch paid ~65 000 page faults
// again (measured on bin/rkmh stream, 16 M reads: 2.4 s of system time against 2.0 s of user time).  Buffers of at least 1 MB
// that this library hands out are remembered with their capacity; rk_seqset_free parks up to POOL_SLOTS of them and the next
// batch's arrays start from a parked buffer instead of a fresh mapping.  rk_free forgets a pointer before freeing it.
namespace pool {
constexpr size_t MIN_BYTES = (size_t)1 << 20, MAX_BYTES = (size_t)512 << 20, POOL_SLOTS = 24, POOL_BYTES = (size_t)3 << 29; // park at most 1.5 GB
struct State {
    std::mutex mu;
    std::unordered_map<void*, size_t> cap;           // every live buffer of >= MIN_BYTES this library handed out
    std::vector<std::pair<size_t, void*>> parked;    // (capacity, buffer) ready for reuse
    size_t parked_bytes = 0;
};
static State& st() { static State* s = new State(); return *s; } // leaked on purpose: buffers may be returned during exit
static void remember(void* p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> g(st().mu);
    // an address this map still knows from an EARLIER buffer (one the caller released with plain free(), which the header allows
    // for malloc'd arrays) must not keep that buffer's capacity: a small buffer at the same address would later be parked and
    // reused with it.  The entry is overwritten, or dropped when the new buffer is too small to be worth parking.
    if (bytes < MIN_BYTES) st().cap.erase(p);
    else st().cap[p] = bytes;
}
static void forget(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(st().mu);
    st().cap.erase(p);
}
// a parked buffer of at least `bytes` (and at most 4x that), or nullptr
static void* take(size_t bytes, size_t* got) {
    if (bytes < MIN_BYTES) return nullptr;
    std::lock_guard<std::mutex> g(st().mu);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < st().parked.size(); ++i)
        if (st().parked[i].first >= bytes && st().parked[i].first <= 4 * bytes && (best == (size_t)-1 || st().parked[i].first < st().parked[best].first)) best = i;
    if (best == (size_t)-1) return nullptr;
    void* p = st().parked[best].second;
    *got = st().parked[best].first;
    st().parked_bytes -= *got;
    st().parked.erase(st().parked.begin() + (long)best);
    return p;
}
// parks p if it is one of ours and there is room; else frees it
static void give_back(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(st().mu);
        auto it = st().cap.find(p);
        if (it != st().cap.end()) {
            const size_t bytes = it->second;
            st().cap.erase(it);
            if (bytes <= MAX_BYTES && st().parked.size() < POOL_SLOTS && st().parked_bytes + bytes <= POOL_BYTES) {
                st().parked.emplace_back(bytes, p);
                st().parked_bytes += bytes;
                return;
            }
        }
    }
    free(p);
}
// frees every parked buffer (a long-lived caller that is done parsing gets its ~1.5 GB back)
static void trim() {
    std::vector<std::pair<size_t, void*>> old;
    {
        std::lock_guard<std::mutex> g(st().mu);
        old.swap(st().parked);
        st().parked_bytes = 0;
    }
    for (auto& e : old) free(e.second);
}
} // namespace pool
extern "C" void rk__pool_forget(void* p) { pool::forget(p); } // rk_free (rk_api.hip) calls this before free()
extern "C" void rk_pool_trim(void) { pool::trim(); }

template <typename V> struct Grow { // malloc-backed growable array handed over to C callers
    V* p = nullptr;
    size_t n = 0, cap = 0;
    // Owns p: movable, never copied.  (It used to have only the implicit, shallow copy: std::vector<Piece>::resize relocating live
    // pieces -- a later block using more workers than an earlier one -- left the new elements pointing at buffers the old ones had
    // freed, and the next reserve() was a double free.  Seen as a rare abort of the parser tests; found with AddressSanitizer.)
    Grow() = default;
    Grow(const Grow&) = delete;
    Grow& operator=(const Grow&) = delete;
    Grow(Grow&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    Grow& operator=(Grow&& o) noexcept {
        if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        if (!p) { // a parked batch buffer of the right size (see pool above)
            size_t got = 0;
            if (void* q = pool::take(want * sizeof(V), &got)) { p = (V*)q; cap = got / sizeof(V); return true; }
        }
        size_t nc = cap ? cap : 1024;
        while (nc < want) nc += nc >> 1;
        V* q = (V*)realloc(p, nc * sizeof(V));
        if (!q) return false;
        p = q; cap = nc;
        return true;
    }
    bool push(V v) { if (n == cap && !reserve(n + 1)) return false; p[n++] = v; return true; }
    bool append(const V* s, size_t k) { if (k == 0) return true; if (!reserve(n + k)) return false; memcpy(p + n, s, k * sizeof(V)); n += k; return true; }
    V* release() { V* r = p; pool::remember(r, cap * sizeof(V)); p = nullptr; n = cap = 0; return r; }
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

struct Piece {
    Grow<uint8_t> bases;
    Grow<uint64_t> ends;      // cumulative base count after each record (piece-local)
    Grow<char> names;
    Grow<uint64_t> name_ends; // cumulative name bytes (with the NULs) after each record
    Grow<char> quals;
    size_t lead = 0;          // FASTA pieces that start inside a record: bases that continue the previous piece's last record
    bool bad = false, oom = false;
};

} // namespace

// Compressed input (one deflate stream: only a sequential reader can follow it, src/rkmh.cpp:238-263): zlib inflates on its OWN
// thread, a few 8 MB pieces ahead, while the block scanner's threads parse what has arrived.
struct AsyncGz {
    struct Piece { std::unique_ptr<unsigned char[]> p; size_t n = 0; }; // (not a vector: no zero fill of megabytes that gzread overwrites)
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Piece> q;
    bool done = false, stop = false;
    Piece cur;
    size_t cur_pos = 0;
};

struct rk_reader {
    gzFile fp = nullptr;
    AsyncGz* az = nullptr;       // compressed input only
    bool compressed = false;
    // a read that FAILED (gzread < 0: a corrupt or truncated gzip stream, an I/O error) -- not an end of input.  Set by whichever
    // thread reads (the inflater thread of compressed input included), reported by rk_reader_next / rk_parse_files: the reference's
    // kseq takes the failed gzread for the end of the file and exits 0 with what it had (src/kseq.hpp:75-85); this build says so
    std::atomic<bool> io_failed{false};
    std::mutex io_mu;
    std::string io_msg;
    // after a gzread that returned r for `want` bytes: r < 0 is a failure, and so is a short read that zlib explains with an error
    // (Z_BUF_ERROR: the compressed stream ends before its end-of-stream mark; Z_DATA_ERROR: damaged)
    void check_read(int r, size_t want) {
        if (r >= 0 && (size_t)r >= want) return;
        int errnum = 0;
        const char* m = gzerror(fp, &errnum);
        if (r >= 0 && (errnum == Z_OK || errnum == Z_STREAM_END)) return;
        std::lock_guard<std::mutex> l(io_mu);
        if (io_failed.load()) return;
        io_msg = m && *m ? m : "read error";
        io_failed.store(true);
    }
    std::string read_error() { std::lock_guard<std::mutex> l(io_mu); return io_msg; }
    // `want` bytes of the (inflated) input, fewer only at its end
    size_t pull(unsigned char* dst, size_t want) {
        if (!compressed) { const int r = gzread(fp, dst, (unsigned)want); check_read(r, want); return r > 0 ? (size_t)r : 0; }
        if (!az) {
            az = new AsyncGz();
            az->th = std::thread([this] {
                size_t want_n = (size_t)256 << 10; // the first pieces are small (a reference panel of a megabyte is one of them), later ones 8 MB
                for (;;) {
                    AsyncGz::Piece piece;
                    piece.p.reset(new unsigned char[want_n]);
                    const int r = gzread(fp, piece.p.get(), (unsigned)want_n);
                    check_read(r, want_n); // (before `done` is published under the lock: the consumer sees it)
                    piece.n = r > 0 ? (size_t)r : 0;
                    const bool last = piece.n < want_n;
                    if (want_n < ((size_t)8 << 20)) want_n *= 4;
                    std::unique_lock<std::mutex> l(az->m);
                    az->cv.wait(l, [&] { return az->q.size() < 6 || az->stop; });
                    if (az->stop) return;
                    if (piece.n) az->q.push_back(std::move(piece));
                    if (last) az->done = true;
                    az->cv.notify_all();
                    if (last) return;
                }
            });
        }
        size_t got = 0;
        while (got < want) {
            if (az->cur_pos == az->cur.n) {
                std::unique_lock<std::mutex> l(az->m);
                az->cv.wait(l, [&] { return !az->q.empty() || az->done; });
                if (az->q.empty()) break; // end of input
                az->cur = std::move(az->q.front());
                az->q.pop_front();
                az->cur_pos = 0;
                az->cv.notify_all();
            }
            const size_t n = std::min(want - got, az->cur.n - az->cur_pos);
            memcpy(dst + got, az->cur.p.get() + az->cur_pos, n);
            az->cur_pos += n; got += n;
        }
        return got;
    }
    std::vector<unsigned char> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    int last_char = 0;
    bool finished = false;

    // --- block-parallel front end (plain, well-formed files only; see next_block) ---
    bool par_ok = false;       // still in block-parallel mode
    bool skip_quals = false;   // RK_READER_NO_QUALS
    int nthreads = 1;
    const unsigned char* map = nullptr; // regular uncompressed file: the whole file, mmap'd
    size_t map_size = 0, map_pos = 0, map_start = 0; // map_start: where this reader began (rk_reader_open_at); map_size: where it ends
    size_t map_total = 0;                            // bytes really mapped (rk_reader_open_range reads a part of the file)
    bool was_irregular = false;                      // the block-parallel front end handed the input to the sequential scanner
    std::vector<unsigned char> blk;     // pipes: current block; starts with the tail of the previous one
    bool blk_eof = false;
    const unsigned char* pend_ptr = nullptr; // after a fallback the sequential scanner is fed from here first
    size_t pend_len = 0, pend_pos = 0;
    bool pending = false, pend_then_eof = false;
    double bytes_per_rec = 0;
    size_t max_block = 384u << 20; // RKMH_PARSE_BLOCK_KB
    size_t min_par = 1u << 20;     // blocks smaller than this are scanned by one thread
    std::vector<Piece> pieces;     // per-worker scratch, reused across blocks

    void to_sequential() {
        if (par_ok) was_irregular = true;
        par_ok = false;
        pending = true;
        pend_pos = 0;
        if (map) { pend_ptr = map + map_pos; pend_len = map_size - map_pos; pend_then_eof = true; }
        else { pend_ptr = blk.data(); pend_len = blk.size(); pend_then_eof = blk_eof; }
    }

    inline bool fill() {
        if (pending) {
            size_t n = pend_len - pend_pos;
            if (n > buf.size()) n = buf.size();
            if (n) {
                memcpy(buf.data(), pend_ptr + pend_pos, n);
                pend_pos += n;
                beg = 0; end = n;
                return true;
            }
            pending = false;
            std::vector<unsigned char>().swap(blk);
            if (pend_then_eof) eof = true;
        }
        if (eof || !fp) return false;
        beg = 0;
        end = pull(buf.data(), buf.size());
        if (end < buf.size()) eof = true;
        return end > 0;
    }
    inline int getc() {
        if (beg >= end && !fill()) return -1;
        return buf[beg++];
    }

    // open-cut parsing restarts the file from byte 0 with the sequential scanner on any irregularity (a block may have
    // begun inside a record, which the sequential scanner cannot resume): the batch's state at the start of the file
    bool f_saved = false, f_quals_ok = true;
    size_t f_nseq = 0, f_bases = 0, f_names = 0, f_quals = 0;
    bool blk_at_ls = true;     // pipes: the byte after the previous block's cut begins a line
    bool rec_open = false;     // a FASTA header has been seen: a block / piece may begin with sequence data of that record
    int next_block(Batch& b, int64_t max_records, bool open_cuts = false);

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
            if (skip_quals) b.quals_ok = false;
            const bool keep_q = b.quals_ok;
            if (keep_q && !b.quals.reserve(q0 + slen + 1)) return -3;
            for (;;) {
                c = getc();
                if (c == -1 || !(ql < slen)) break;
                if (c >= 33 && c <= 127) { if (keep_q) b.quals.p[q0 + ql] = (char)c; ++ql; }
            }
            last_char = 0;
            if (ql != slen) { b.bases.n = seq0; b.names.n = name0; return -2; }
            if (keep_q) b.quals.n = q0 + slen;
        } else {
            b.quals_ok = false;
        }
        if (!b.names.push('\0')) return -3;
        if (!b.offsets.push((uint64_t)b.bases.n) || !b.name_offsets.push((uint64_t)b.names.n)) return -3;
        ++b.nseq;
        return 1;
    }
};

// ---------------------------------------------------------------------------------------------------
// Block-parallel front end.  kseq's grammar is inherently sequential (a quality string is read by count,
// '@' may be data), so the parallel path only accepts input on which that grammar provably coincides with
// the line-oriented one: every FASTQ record is exactly four lines (header, one sequence line of keeper
// bytes, a '+' line, a quality line of exactly as many bytes 33..127), every FASTA sequence line holds only
// keeper bytes.  A block is cut at validated record starts, the pieces are scanned by worker threads under
// those strict rules, and ANY deviation anywhere in the block hands the untouched block back to the
// sequential scanner above (and disables this path for the rest of the file), so the result is always the
// one kseq_read (kseq.hpp:170-208) would produce.
// ---------------------------------------------------------------------------------------------------
namespace {

inline const unsigned char* find_nl(const unsigned char* p, const unsigned char* e) {
    return p < e ? (const unsigned char*)memchr(p, '\n', (size_t)(e - p)) : nullptr;
}

// true when [p, e) are all keeper bytes (class 1)
inline bool all_keepers(const unsigned char* p, const unsigned char* e) {
    unsigned acc = 0; // arithmetic form of seqcls == 1 so the loop vectorises
    for (; p < e; ++p) {
        const unsigned c = *p;
        acc |= (unsigned)((c - 33u) > 93u) | (unsigned)(c == '>') | (unsigned)(c == '+') | (unsigned)(c == '@');
    }
    return acc == 0;
}
inline bool all_quals(const unsigned char* p, const unsigned char* e) {
    unsigned acc = 0;
    for (; p < e; ++p) acc |= (unsigned)((unsigned)(*p - 33u) > 94u);
    return acc == 0;
}

inline bool push_name(Piece& o, const unsigned char* p, const unsigned char* nl) {
    const unsigned char* q = p;
    while (q < nl && !T.space[*q]) ++q;
    if (!o.names.append((const char*)p, (size_t)(q - p)) || !o.names.push('\0')) return false;
    return o.name_ends.push((uint64_t)o.names.n);
}

void parse_fastq_piece(const unsigned char* p, const unsigned char* e, bool want_quals, Piece& o) {
    while (p < e) {
        while (p < e && (*p == '\n' || *p == '\r')) ++p;
        if (p >= e) break;
        if (*p != '@') { o.bad = true; return; }
        ++p;
        const unsigned char* nl = find_nl(p, e);
        if (!nl) { o.bad = true; return; }
        if (!push_name(o, p, nl)) { o.oom = true; return; }
        p = nl + 1;
        nl = find_nl(p, e);
        if (!nl) { o.bad = true; return; }
        const unsigned char* se = nl;
        if (se > p && se[-1] == '\r') --se;
        const size_t slen = (size_t)(se - p);
        if (!all_keepers(p, se)) { o.bad = true; return; }
        if (!o.bases.append(p, slen) || !o.ends.push((uint64_t)o.bases.n)) { o.oom = true; return; }
        p = nl + 1;
        if (p >= e || *p != '+') { o.bad = true; return; }
        nl = find_nl(p, e);
        if (!nl) { o.bad = true; return; }
        p = nl + 1;
        nl = find_nl(p, e);
        const unsigned char* qe = nl ? nl : e;
        const unsigned char* qend = qe;
        if (qend > p && qend[-1] == '\r') --qend;
        if ((size_t)(qend - p) != slen || !all_quals(p, qend)) { o.bad = true; return; }
        if (want_quals && !o.quals.append((const char*)p, slen)) { o.oom = true; return; }
        p = nl ? nl + 1 : e;
    }
}

// FASTA piece [p, e).  A piece may begin INSIDE a record (cuts may fall anywhere outside a header line, so that a
// chromosome-sized record is shared by all workers): bases before the piece's first header line are its `lead`, the
// continuation of the record that was open when the piece began.  at_ls: p is the first byte of a line.  need_header:
// nothing is open yet (start of the file), so anything but blank lines before the first header is an irregularity.
void parse_fasta_piece(const unsigned char* p, const unsigned char* e, bool at_ls, bool need_header, Piece& o) {
    bool opened = false; // a header was seen in this piece
    o.lead = 0;
    while (p < e) {
        if (at_ls && *p == '>') {
            if (opened) { if (!o.ends.push((uint64_t)o.bases.n)) { o.oom = true; return; } }
            else o.lead = o.bases.n;
            ++p;
            const unsigned char* nl = find_nl(p, e);
            if (!nl) { o.bad = true; return; } // the file ends inside a header line: leave it to the sequential scanner
            if (!push_name(o, p, nl)) { o.oom = true; return; }
            p = nl + 1;
            opened = true;
            continue;
        }
        if (!opened && need_header) { // before the very first header only blank lines are regular
            if (*p == '\n' || *p == '\r') { at_ls = *p == '\n'; ++p; continue; }
            o.bad = true; return;
        }
        const unsigned char* nl = find_nl(p, e);
        const unsigned char* se = nl ? nl : e;
        if (nl && se > p && se[-1] == '\r') --se;
        if (!all_keepers(p, se)) { o.bad = true; return; }
        if (!o.bases.append(p, (size_t)(se - p))) { o.oom = true; return; }
        p = nl ? nl + 1 : e;
        at_ls = nl != nullptr;
    }
    if (opened) { if (!o.ends.push((uint64_t)o.bases.n)) { o.oom = true; return; } }
    else o.lead = o.bases.n;
}

// a legal FASTA cut at or before `pos`: anywhere except inside a header line (then the start of that line)
size_t fasta_cut(const unsigned char* base, size_t pos) {
    if (pos == 0) return 0;
    const unsigned char* nl = (const unsigned char*)memrchr(base, '\n', pos);
    const size_t ls = nl ? (size_t)(nl - base) + 1 : 0;
    return base[ls] == '>' ? ls : pos;
}

// first record start at or after `from` (line-aligned); for FASTQ the line two below must begin with '+',
// which a quality line that happens to begin with '@' can never satisfy (its +2 line is a sequence line).
const unsigned char* find_record_start(const unsigned char* base, const unsigned char* from,
                                       const unsigned char* e, bool fastq) {
    const unsigned char* p = from;
    if (p > base && p[-1] != '\n') {
        const unsigned char* nl = find_nl(p, e);
        if (!nl) return nullptr;
        p = nl + 1;
    }
    while (p < e) {
        if (fastq) {
            if (*p == '@') {
                const unsigned char* l1 = find_nl(p, e);
                if (!l1) return nullptr;
                const unsigned char* l2 = find_nl(l1 + 1, e);
                if (!l2 || l2 + 1 >= e) return nullptr;
                if (l2[1] == '+') return p;
            }
        } else if (*p == '>') {
            return p;
        }
        const unsigned char* nl = find_nl(p, e);
        if (!nl) return nullptr;
        p = nl + 1;
    }
    return nullptr;
}

// last record start in (base, e): records before it are complete inside the block
const unsigned char* find_last_record_start(const unsigned char* base, const unsigned char* e, bool fastq) {
    size_t win = 1u << 16;
    const size_t total = (size_t)(e - base);
    for (;;) {
        const unsigned char* from = total > win ? e - win : base + 1;
        const unsigned char* best = nullptr;
        const unsigned char* p = from;
        for (;;) {
            const unsigned char* q = find_record_start(base, p, e, fastq);
            if (!q) break;
            best = q;
            p = q + 1;
        }
        if (best && best > base) return best;
        if (win >= total) return nullptr;
        win *= 16;
    }
}

} // namespace

// Appends the records of the next block to b.  1 = block consumed (more may follow), 0 = end of input,
// -1 = input is not strictly line-structured: the block was handed back to the sequential scanner,
// -3 = out of memory.
int rk_reader::next_block(Batch& b, int64_t max_records, bool open_cuts) {
    // open_cuts (whole-file parsing of a mapped file into ONE batch): FASTA blocks and pieces may end inside a record
    if (!map) open_cuts = false;
    if (open_cuts && !f_saved) {
        f_saved = true; f_quals_ok = b.quals_ok;
        f_nseq = (size_t)b.nseq; f_bases = b.bases.n; f_names = b.names.n; f_quals = b.quals.n;
    }
    auto irregular = [&]() -> int { // hand the input to the sequential scanner
        if (open_cuts) { // ... from the first byte of the file, dropping what this file contributed so far
            b.nseq = (int64_t)f_nseq; b.bases.n = f_bases; b.names.n = f_names; b.quals.n = f_quals; b.quals_ok = f_quals_ok;
            b.offsets.n = f_nseq + 1; b.name_offsets.n = f_nseq + 1;
            map_pos = map_start; rec_open = false;
        }
        to_sequential();
        return -1;
    };
    size_t target = bytes_per_rec > 0 && max_records > 0 ? (size_t)(bytes_per_rec * (double)max_records) : (64u << 20);
    if (open_cuts) target = max_block;
    if (target < (4u << 20)) target = 4u << 20;
    if (target > max_block) target = max_block;
    bool fasta_open = false; // this block uses the open-record FASTA rules
    bool fastq = false;
    size_t cut = 0;
    const unsigned char* base = nullptr;
    for (;;) {
        size_t avail;
        bool at_eof;
        if (map) {
            avail = map_size - map_pos;
            at_eof = avail <= target;
            if (!at_eof) avail = target;
            base = map + map_pos;
        } else {
            while (!blk_eof && blk.size() < target) { // top the block up to `target` bytes
                size_t have = blk.size();
                size_t want = target - have;
                if (want > (1u << 30)) want = 1u << 30;
                blk.resize(have + want);
                size_t got = pull(blk.data() + have, want);
                blk.resize(have + got);
                if (got < want) blk_eof = true;
            }
            avail = blk.size();
            at_eof = blk_eof;
            base = blk.data();
        }
        if (avail == 0) return 0;
        const unsigned char* e = base + avail;
        const unsigned char* p = base;
        while (p < e && (*p == '\n' || *p == '\r')) ++p;
        if (p >= e) {
            if (at_eof) { if (map) map_pos = map_size; else blk.clear(); return 0; }
            target *= 2;
            continue;
        }
        if (open_cuts && rec_open) { // continuing a FASTA file: the block may begin with sequence data
            fastq = false; fasta_open = true;
            cut = at_eof ? avail : fasta_cut(base, avail);
            if (cut > 0) break;
            target *= 2; // a header line longer than the block
            continue;
        }
        if (*p != '@' && *p != '>') return irregular();
        fastq = *p == '@';
        if (!fastq && open_cuts) {
            fasta_open = true;
            cut = at_eof ? avail : fasta_cut(base, avail);
            if (cut > 0) break;
            target *= 2;
            continue;
        }
        if (at_eof) { cut = avail; break; }
        const unsigned char* last = find_last_record_start(base, e, fastq);
        if (last) { cut = (size_t)(last - base); break; }
        target *= 2; // a single record longer than the block: look further
    }
    int nt = nthreads;
    if (cut < min_par) nt = 1;
    std::vector<size_t> starts((size_t)nt + 1);
    starts[0] = 0;
    starts[(size_t)nt] = cut;
    for (int i = 1; i < nt; ++i) {
        size_t from = cut / (size_t)nt * (size_t)i;
        if (from < starts[(size_t)i - 1]) from = starts[(size_t)i - 1];
        if (fasta_open) {
            size_t c = fasta_cut(base, from);
            starts[(size_t)i] = c < starts[(size_t)i - 1] ? starts[(size_t)i - 1] : c;
            continue;
        }
        const unsigned char* q = find_record_start(base, base + from, base + cut, fastq);
        starts[(size_t)i] = q ? (size_t)(q - base) : cut;
    }
    if (pieces.size() < (size_t)nt) pieces.resize((size_t)nt);
    const bool want_quals = fastq && b.quals_ok && !skip_quals;
    auto work = [&](int i) {
        Piece& pc = pieces[(size_t)i];
        pc.bases.n = pc.ends.n = pc.names.n = pc.name_ends.n = pc.quals.n = 0;
        pc.bad = pc.oom = false;
        const unsigned char* s = base + starts[(size_t)i];
        const unsigned char* t = base + starts[(size_t)i + 1];
        // upper bounds up front (untouched pages cost nothing): no geometric regrowth copies of 100 MB sequence lines
        const size_t span = (size_t)(t - s);
        if (!pc.bases.reserve(fastq ? span / 2 + 64 : span + 64) || (want_quals && !pc.quals.reserve(span / 2 + 64))) { pc.oom = true; return; }
        if (fastq) parse_fastq_piece(s, t, want_quals, pc);
        else {
            const bool at_ls = s == base ? (fasta_open ? (map ? (map_pos == 0 || s[-1] == '\n') : blk_at_ls) : true) : s[-1] == '\n';
            parse_fasta_piece(s, t, at_ls, /*need_header=*/i == 0 && !(fasta_open && rec_open), pc);
            if (!fasta_open && pc.lead) pc.bad = true; // closed-cut blocks begin at a record start by construction
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < nt; ++i) th.emplace_back(work, i);
        work(0);
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < nt; ++i) {
        if (pieces[(size_t)i].oom) return -3;
        if (pieces[(size_t)i].bad) return irregular();
    }
    // merge: prefix sums, then every worker copies its own piece into place
    std::vector<size_t> rec0((size_t)nt + 1), base0((size_t)nt + 1), name0((size_t)nt + 1);
    rec0[0] = (size_t)b.nseq; base0[0] = b.bases.n; name0[0] = b.names.n;
    for (int i = 0; i < nt; ++i) {
        rec0[(size_t)i + 1] = rec0[(size_t)i] + pieces[(size_t)i].ends.n;
        base0[(size_t)i + 1] = base0[(size_t)i] + pieces[(size_t)i].bases.n;
        name0[(size_t)i + 1] = name0[(size_t)i] + pieces[(size_t)i].names.n;
    }
    const size_t nrec = rec0[(size_t)nt], nb = base0[(size_t)nt], nn = name0[(size_t)nt];
    const size_t q_before = b.quals.n;
    if (!b.bases.reserve(nb + 64) || !b.offsets.reserve(nrec + 1) || !b.names.reserve(nn + 1) ||
        !b.name_offsets.reserve(nrec + 1))
        return -3;
    if (want_quals && !b.quals.reserve(q_before + (nb - base0[0]) + 1)) return -3;
    auto place = [&](int i) {
        Piece& pc = pieces[(size_t)i];
        if (pc.bases.n) memcpy(b.bases.p + base0[(size_t)i], pc.bases.p, pc.bases.n);
        if (pc.names.n) memcpy(b.names.p + name0[(size_t)i], pc.names.p, pc.names.n);
        if (want_quals && pc.quals.n)
            memcpy(b.quals.p + q_before + (base0[(size_t)i] - base0[0]), pc.quals.p, pc.quals.n);
        uint64_t* off = b.offsets.p + rec0[(size_t)i] + 1;
        uint64_t* noff = b.name_offsets.p + rec0[(size_t)i] + 1;
        for (size_t j = 0; j < pc.ends.n; ++j) {
            off[j] = (uint64_t)base0[(size_t)i] + pc.ends.p[j];
            noff[j] = (uint64_t)name0[(size_t)i] + pc.name_ends.p[j];
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < nt; ++i) th.emplace_back(place, i);
        place(0);
        for (auto& t : th) t.join();
    }
    if (fasta_open) { // a piece's leading bases belong to the record that was open when the piece began
        for (int i = 0; i < nt; ++i) {
            const size_t lead = pieces[(size_t)i].lead;
            if (!lead) continue;
            if (rec0[(size_t)i] == 0) return irregular(); // sequence data before any header (cannot happen: need_header)
            b.offsets.p[rec0[(size_t)i]] = (uint64_t)base0[(size_t)i] + lead;
        }
        if (nrec > 0) rec_open = true;
        if (!map) blk_at_ls = cut > 0 && base[cut - 1] == '\n';
    }
    b.bases.n = nb; b.names.n = nn;
    b.offsets.n = nrec + 1; b.name_offsets.n = nrec + 1;
    if (want_quals) b.quals.n = q_before + (nb - base0[0]);
    const size_t got = nrec - (size_t)b.nseq;
    if (got && !want_quals) b.quals_ok = false;
    if (got) bytes_per_rec = (double)cut / (double)got;
    b.nseq = (int64_t)nrec;
    if (map) {
        map_pos += cut;
    } else { // keep the unfinished tail for the next call
        const size_t tail = blk.size() - cut;
        if (tail) memmove(blk.data(), blk.data() + cut, tail);
        blk.resize(tail);
    }
    return 1;
}

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

// CPUs this process may actually use: the affinity mask and the cgroup quota (a 256-thread host that grants a container 16 CPUs
// reports 256 from hardware_concurrency)
static int granted_cpus() {
    static const int n = [] {
        long v = (long)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const long a = CPU_COUNT(&set); if (a > 0 && (v <= 0 || a < v)) v = a; }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long c = (atol(q) + period - 1) / period;
                if (c > 0 && (v <= 0 || c < v)) v = c;
            }
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
            long quota = -1, period = 0;
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 0; fclose(h); }
            if (quota > 0 && period > 0) { const long c = (quota + period - 1) / period; if (v <= 0 || c < v) v = c; }
        }
        return (int)(v > 0 ? v : 8);
    }();
    return n;
}

int64_t rk_fastq_cut(const uint8_t* text, uint64_t n) {
    if (!text || n < 2) return -1;
    const unsigned char* p = find_last_record_start(text, text + n, true);
    return p ? (int64_t)(p - text) : -1;
}

static int reader_open_at(const char* path, uint64_t offset, rk_reader** out);
int rk_reader_open(const char* path, rk_reader** out) { return reader_open_at(path, 0, out); }
// A reader that starts `offset` bytes into a regular uncompressed file, at a record start (the device-side FASTQ front end hands
// a file over to this scanner at the first block it refuses).
int rk_reader_open_at(const char* path, uint64_t offset, rk_reader** out) {
    if (path && strcmp(path, "-") == 0 && offset) return perr(RK_ERR_ARG, "rk_reader_open_at: STDIN has no offsets");
    return reader_open_at(path, offset, out);
}
// The records that START in bytes [lo, hi) of a regular uncompressed FASTQ file: from the first record start at or after lo (the
// four-line rule of find_record_start; lo = 0: the file's first byte) up to the first record start at or after hi.  Ranks of a
// multi-process run each read their own range this way.  The rule is only right for text that IS four lines per record, so the
// caller must check rk_reader_strict() after the last batch: 0 there (or here: RK_ERR_ARG for input that cannot be split at all)
// means "parse the whole file instead".
int rk_reader_open_range(const char* path, uint64_t lo, uint64_t hi, rk_reader** out) {
    if (!path || !out || hi < lo || strcmp(path, "-") == 0) return perr(RK_ERR_ARG, "bad arguments");
    rk_reader* r = nullptr;
    int rc = reader_open_at(path, 0, &r);
    if (rc != RK_OK) return rc;
    if (!r->map || !r->par_ok || r->map_size == 0 || r->map[0] != '@') { rk_reader_close(r); return perr(RK_ERR_ARG, "byte ranges need an uncompressed regular FASTQ file"); }
    const unsigned char* base = r->map;
    const unsigned char* end = base + r->map_size;
    auto start_at = [&](uint64_t off) -> size_t {
        if (off == 0) return 0;
        if (off >= r->map_size) return r->map_size;
        const unsigned char* p = find_record_start(base, base + off, end, true);
        return p ? (size_t)(p - base) : r->map_size;
    };
    const size_t s = start_at(lo), e = start_at(hi);
    r->map_start = r->map_pos = s;
    r->map_size = e < s ? s : e;
    *out = r;
    return RK_OK;
}
// 1 while everything this reader has returned was read by the strict block-parallel front end (text that is four lines per
// record, on which its cuts and rk_reader_open_range's boundaries are exact); 0 once the sequential kseq scanner took over
int rk_reader_strict(const rk_reader* r) { return r && r->par_ok && !r->was_irregular ? 1 : 0; }

static int reader_open_at(const char* path, uint64_t offset, rk_reader** out) {
    if (!path || !out) return perr(RK_ERR_ARG, "bad arguments");
    rk_reader* r = new rk_reader();
    r->buf.resize(4u << 20);
    {
        const char* e = getenv("RKMH_PARSE_THREADS");
        long v = e ? atol(e) : 0;
        // default: three quarters of the CPUs granted, 4 .. 16 (the classify and formatting threads of a stream run need the
        // rest; measured on 16 CPUs: 16 M reads' main loop 0.38 s with 8 parser threads, 0.32 s with 12, 0.34 s with 16)
        const long dflt = std::min<long>(16, std::max<long>(4, (long)granted_cpus() * 3 / 4));
        r->nthreads = (int)(v > 0 ? (v > 64 ? 64 : v) : dflt);
    }
    if (const char* e = getenv("RKMH_PARSE_BLOCK_KB")) { // testing knob: small blocks exercise cut/carry/merge
        long v = atol(e);
        if (v > 0) { r->max_block = (size_t)v << 10; r->min_par = r->max_block / 8; }
    }
    bool plain = false;
    if (strcmp(path, "-") != 0) { // regular uncompressed file: map it, no read() copies at all
        int fd = open(path, O_RDONLY);
        if (fd < 0) { delete r; return perr(RK_ERR_IO, std::string("cannot open ") + path); }
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) >= 1 &&
            !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                r->map = (const unsigned char*)m;
                r->map_size = r->map_total = (size_t)st.st_size;
                r->map_start = r->map_pos = (size_t)std::min<uint64_t>(offset, (uint64_t)st.st_size);
                plain = true;
            }
        }
        close(fd);
    }
    if (!r->map) {
        gzFile fp = strcmp(path, "-") == 0 ? gzdopen(0, "r") : gzopen(path, "r");
        if (!fp) { delete r; return perr(RK_ERR_IO, std::string("cannot open ") + path); }
        gzbuffer(fp, 1 << 20);
        r->fp = fp;
        plain = gzdirect(fp) != 0;
        // compressed: the inflater gets its own thread, and the block-parallel scanner parses what it delivers (RKMH_GZ_BLOCKS=0: the
        // sequential scanner on the calling thread, as before round 5)
        r->compressed = !plain && !(getenv("RKMH_GZ_BLOCKS") && atoi(getenv("RKMH_GZ_BLOCKS")) == 0);
        if (r->compressed && strcmp(path, "-") != 0) { // a small file (a reference panel) is done before a thread and a block would be set up
            struct stat zst;
            if (stat(path, &zst) == 0 && zst.st_size < ((off_t)8 << 20) && !(getenv("RKMH_GZ_BLOCKS") && atoi(getenv("RKMH_GZ_BLOCKS")) == 2)) r->compressed = false;
        }
        if (r->compressed) plain = true;
        if (offset && gzseek(fp, (z_off_t)offset, SEEK_SET) < 0) { gzclose(fp); delete r; return perr(RK_ERR_IO, std::string("cannot seek in ") + path); }
    }
    r->par_ok = plain && r->nthreads > 1;
    if (!r->par_ok && r->map) r->to_sequential();
    *out = r;
    return RK_OK;
}

void rk_reader_set_options(rk_reader* r, int flags) {
    if (r) r->skip_quals = (flags & RK_READER_NO_QUALS) != 0;
}

void rk_reader_close(rk_reader* r) {
    if (!r) return;
    if (r->az) {
        { std::lock_guard<std::mutex> l(r->az->m); r->az->stop = true; }
        r->az->cv.notify_all();
        if (r->az->th.joinable()) r->az->th.join();
        delete r->az;
    }
    if (r->fp) gzclose(r->fp);
    if (r->map) munmap((void*)r->map, r->map_total);
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
        if (r->par_ok && max_records > 0 && max_records < 65536) r->to_sequential(); // small exact batches
        if (r->par_ok) { // block-parallel: limits are soft (a whole block is appended at a time)
            int rb = r->next_block(b, max_records);
            if (rb == -3) return perr(RK_ERR_NOMEM, "out of memory");
            if (rb == 0) { r->finished = true; break; }
            if (rb == 1) break;
            continue; // -1: fell back to the sequential scanner
        }
        int rc = r->next(b);
        if (rc == -3) return perr(RK_ERR_NOMEM, "out of memory");
        if (rc <= 0) { r->finished = true; break; } // EOF, or truncated record: ends the file (rkmh.cpp:251)
    }
    if (r->io_failed.load()) return perr(RK_ERR_IO, ("read failed (corrupt or truncated input?): " + r->read_error()).c_str());
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
            int k = r->par_ok ? r->next_block(b, 0, /*open_cuts=*/true) : r->next(b);
            if (k == -3) { rk_reader_close(r); return perr(RK_ERR_NOMEM, "out of memory"); }
            if (k == -1) continue;
            if (k <= 0) break;
        }
        if (r->io_failed.load()) {
            const std::string msg = std::string(paths[i]) + ": read failed (corrupt or truncated input?): " + r->io_msg;
            rk_reader_close(r);
            return perr(RK_ERR_IO, msg.c_str());
        }
        rk_reader_close(r);
    }
    return hand_over(b, out);
}

void rk_seqset_free(rk_seqset* s) {
    if (!s) return;
    pool::give_back(s->bases); pool::give_back(s->offsets); pool::give_back(s->names); pool::give_back(s->name_offsets); pool::give_back(s->quals);
    memset(s, 0, sizeof *s);
}


} // extern "C"

// ------------------------------------------------------------------------------------------------------------------------
// BGZF (bgzip) input: the file is a chain of INDEPENDENT gzip members of at most 64 KB of text each, and every member's header
// says how long the member is ('BC' extra field) -- so, unlike plain gzip (one deflate stream that only a sequential reader can
// follow, src/rkmh.cpp:238-263 through gzFile), the members can be found without inflating anything and inflated by as many
// threads as there are.  The device FASTQ front end's workers each inflate the members of their job straight in front of the
// upload (rkmh_main.cpp, stream_file_raw); nothing in the process ever reads the file sequentially.
// Inflate = libdeflate when the system has it (dlopen: ~3 x zlib's rate per core), zlib otherwise; CRC-32 and ISIZE are checked.
struct rk_bgzf {
    int fd = -1;
    const unsigned char* map = nullptr;
    size_t size = 0;
    std::vector<uint64_t> coff, uoff; // [members + 1] offset in the file / in the text
    std::vector<uint16_t> hlen;       // [members] header bytes (12 + XLEN)
};

namespace {
struct Deflate {
    void* h = nullptr;
    void* (*alloc)() = nullptr;
    int (*run)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    Deflate() {
        if (getenv("RKMH_NO_LIBDEFLATE")) return;
        h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        run = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!alloc || !run || !crc) { alloc = nullptr; run = nullptr; crc = nullptr; }
    }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }

// one member's text into dst[0 .. usize); false: corrupt
bool inflate_member(const rk_bgzf* z, size_t b, unsigned char* dst) {
    const unsigned char* m = z->map + z->coff[b];
    const size_t total = (size_t)(z->coff[b + 1] - z->coff[b]), h = z->hlen[b], usize = (size_t)(z->uoff[b + 1] - z->uoff[b]);
    const unsigned char* payload = m + h;
    const size_t clen = total - h - 8;
    uint32_t want_crc;
    memcpy(&want_crc, m + total - 8, 4);
    const Deflate& L = deflate_lib();
    if (L.run) {
        static thread_local void* dec = L.alloc();
        size_t got = 0;
        if (!dec || L.run(dec, payload, clen, dst, usize, &got) != 0 || got != usize) return false;
        return L.crc(0, dst, usize) == want_crc;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char*>(payload); zs.avail_in = (uInt)clen;
    zs.next_out = dst; zs.avail_out = (uInt)usize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == usize;
    inflateEnd(&zs);
    return ok && (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, (uInt)usize) == want_crc;
}
} // namespace

extern "C" {

// RK_OK and *out = the opened file; RK_ERR_ARG: not a BGZF file (not an error to report: the caller takes another path)
int rk_bgzf_open(const char* path, rk_bgzf** out) {
    if (!path || !out) return perr(RK_ERR_ARG, "bad arguments");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return perr(RK_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 28) { close(fd); return perr(RK_ERR_ARG, "not a BGZF file"); }
    void* mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (mp == MAP_FAILED) { close(fd); return perr(RK_ERR_IO, std::string("cannot map ") + path); }
    rk_bgzf* z = new rk_bgzf();
    z->fd = fd; z->map = (const unsigned char*)mp; z->size = (size_t)st.st_size;
    size_t p = 0;
    uint64_t u = 0;
    bool ok = true;
    while (ok && p < z->size) {
        const unsigned char* m = z->map + p;
        if (z->size - p < 26 || m[0] != 0x1f || m[1] != 0x8b || m[2] != 8 || m[3] != 4) { ok = false; break; } // FLG = FEXTRA alone, as bgzip writes it
        const size_t xlen = (size_t)m[10] | ((size_t)m[11] << 8);
        if (12 + xlen + 8 > z->size - p) { ok = false; break; }
        size_t bsize = 0;
        for (size_t q = 12; q + 4 <= 12 + xlen;) {
            const size_t slen = (size_t)m[q + 2] | ((size_t)m[q + 3] << 8);
            if (m[q] == 'B' && m[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = ((size_t)m[q + 4] | ((size_t)m[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > z->size - p) { ok = false; break; }
        uint32_t isize;
        memcpy(&isize, m + bsize - 4, 4);
        if (isize > 65536) { ok = false; break; }
        z->coff.push_back(p); z->uoff.push_back(u); z->hlen.push_back((uint16_t)(12 + xlen));
        p += bsize; u += isize;
    }
    if (!ok || z->coff.empty()) { rk_bgzf_close(z); return perr(RK_ERR_ARG, "not a BGZF file"); }
    z->coff.push_back(p); z->uoff.push_back(u);
    madvise(mp, z->size, MADV_WILLNEED);
    *out = z;
    return RK_OK;
}
void rk_bgzf_close(rk_bgzf* z) {
    if (!z) return;
    if (z->map) munmap((void*)z->map, z->size);
    if (z->fd >= 0) close(z->fd);
    delete z;
}
int64_t rk_bgzf_members(const rk_bgzf* z) { return z ? (int64_t)z->hlen.size() : 0; }
// the file image and one member's place in it: bytes [*file_off, *file_off + *total) = header (*header bytes) + deflate stream + CRC-32 + ISIZE
const uint8_t* rk_bgzf_image(const rk_bgzf* z) { return z ? z->map : nullptr; }
uint64_t rk_bgzf_file_bytes(const rk_bgzf* z) { return z ? (uint64_t)z->size : 0; }
int rk_bgzf_member(const rk_bgzf* z, int64_t m, uint64_t* file_off, uint32_t* total, uint32_t* header, uint32_t* text_bytes) {
    if (!z || m < 0 || (size_t)m >= z->hlen.size()) return perr(RK_ERR_ARG, "bad arguments");
    if (file_off) *file_off = z->coff[(size_t)m];
    if (total) *total = (uint32_t)(z->coff[(size_t)m + 1] - z->coff[(size_t)m]);
    if (header) *header = z->hlen[(size_t)m];
    if (text_bytes) *text_bytes = (uint32_t)(z->uoff[(size_t)m + 1] - z->uoff[(size_t)m]);
    return RK_OK;
}
uint64_t rk_bgzf_text_bytes(const rk_bgzf* z) { return z ? z->uoff.back() : 0; }
uint64_t rk_bgzf_text_offset(const rk_bgzf* z, int64_t member) {
    if (!z || member < 0) return 0;
    return z->uoff[(size_t)std::min<int64_t>(member, (int64_t)z->hlen.size())];
}
// the text's first byte (0: empty or corrupt): '@' = FASTQ
int rk_bgzf_first_byte(const rk_bgzf* z) {
    if (!z) return 0;
    std::vector<unsigned char> t(65536);
    for (size_t b = 0; b < z->hlen.size(); ++b) {
        if (z->uoff[b + 1] == z->uoff[b]) continue;
        return inflate_member(z, b, t.data()) ? (int)t[0] : 0;
    }
    return 0;
}
// consecutive members grouped into jobs of about target_bytes of text: first[j] = first member of job j, first[jobs] = members
int64_t rk_bgzf_plan(const rk_bgzf* z, uint64_t target_bytes, int64_t* first, int64_t cap) { return rk_bgzf_plan_members(z, target_bytes, INT64_MAX, first, cap); }
// ... and of at most max_members members each (the device inflater decodes 64 members per wave: a job of 16 384 members is 256
// waves, two of which fill the chip -- one member more and a launch has a straggler that waits for the other launch's slots)
int64_t rk_bgzf_plan_members(const rk_bgzf* z, uint64_t target_bytes, int64_t max_members, int64_t* first, int64_t cap) {
    if (!z || !first || cap < 2 || max_members < 1) return perr(RK_ERR_ARG, "bad arguments");
    const size_t nb = z->hlen.size();
    int64_t nj = 0;
    size_t b = 0;
    while (b < nb) {
        if (nj + 1 >= cap) return perr(RK_ERR_LIMIT, "rk_bgzf_plan: more jobs than the caller's array holds");
        first[nj++] = (int64_t)b;
        const uint64_t start = z->uoff[b];
        const size_t b_first = b;
        while (b < nb && z->uoff[b + 1] - start <= target_bytes && (int64_t)(b - b_first) < max_members) ++b;
        if (b == b_first && b < nb) ++b; // (a member larger than the target: alone)
    }
    first[nj] = (int64_t)nb;
    return nj;
}
// The member that holds the byte in front of member b0's text: b0 - 1, unless that one is empty (the end-of-file marker `cat a.gz
// b.gz` leaves in the middle of a file) -- then the nearest one before it with text; b0 itself when no text precedes it.  Both
// record cutters (here and rk_inflate.hip's) begin their text there, so the cut at b0 sees the real previous byte and agrees with
// the cut the previous job made at its end.
int64_t rk_bgzf_lead_member(const rk_bgzf* z, int64_t b0) {
    if (!z || b0 <= 0) return 0;
    if ((size_t)b0 > z->hlen.size()) b0 = (int64_t)z->hlen.size();
    int64_t m = b0 - 1;
    while (m > 0 && z->uoff[(size_t)m + 1] == z->uoff[(size_t)m]) --m;
    return z->uoff[(size_t)m + 1] == z->uoff[(size_t)m] ? b0 : m;
}
// The whole FASTQ records that START in the text of members [b0, b1): the members are inflated (with the one in front, for its
// last byte, and as many behind as the last record reaches into), the first record start at or after the text of b0 and of b1
// is found by the four-line rule of find_record_start -- the same function at both ends, so neighbouring jobs agree -- and the
// bytes between them are copied to dst.  *text_off = where they begin in the uncompressed text.  Returns RK_OK, RK_ERR_IO
// (corrupt member), RK_ERR_LIMIT (more than cap bytes), or 1: the text does not begin with '@' (member 0 only).
int rk_bgzf_fastq_records(const rk_bgzf* z, int64_t b0, int64_t b1, uint8_t* dst, uint64_t cap, uint64_t* nbytes, uint64_t* text_off) {
    if (!z || !dst || !nbytes || b0 < 0 || b1 < b0 || (size_t)b1 > z->hlen.size()) return perr(RK_ERR_ARG, "bad arguments");
    *nbytes = 0;
    const size_t nb = z->hlen.size();
    const size_t lo = (size_t)rk_bgzf_lead_member(z, b0);
    static thread_local std::vector<unsigned char> buf;
    size_t have_b = lo; // members [lo, have_b) are in buf
    auto extend_to = [&](size_t upto) -> bool {
        if (upto > nb) upto = nb;
        const size_t need = (size_t)(z->uoff[upto] - z->uoff[lo]);
        if (buf.size() < need + 64) buf.resize(need + (need >> 2) + 65536);
        for (; have_b < upto; ++have_b)
            if (!inflate_member(z, have_b, buf.data() + (z->uoff[have_b] - z->uoff[lo]))) return false;
        return true;
    };
    if (!extend_to((size_t)b1)) return perr(RK_ERR_IO, "corrupt BGZF member");
    const uint64_t u_lo = z->uoff[lo];
    auto first_start = [&](size_t member, bool* need_more) -> size_t { // offset in buf; == text in buf when there is none
        *need_more = false;
        const unsigned char* base = buf.data();
        const unsigned char* e = base + (z->uoff[have_b] - u_lo);
        const unsigned char* from = base + (z->uoff[member] - u_lo);
        if (z->uoff[member] == 0) return 0; // (no text in front of it: the file's first record)
        if (from >= e) { *need_more = have_b < nb; return (size_t)(e - base); }
        const unsigned char* q = find_record_start(base, from, e, true);
        if (q) return (size_t)(q - base);
        *need_more = have_b < nb;
        return (size_t)(e - base);
    };
    bool more = false;
    size_t tail = first_start((size_t)b1, &more);
    for (size_t step = 2; more; step *= 2) { // the last record reaches into the members behind
        if (!extend_to(have_b + step)) return perr(RK_ERR_IO, "corrupt BGZF member");
        tail = first_start((size_t)b1, &more);
    }
    size_t head = first_start((size_t)b0, &more);
    if (head > tail) head = tail;
    if (z->uoff[(size_t)b0] == 0 && tail > 0 && buf[0] != '@') return 1; // (the job begins the file's text)
    const size_t n = tail - head;
    if (text_off) *text_off = u_lo + head;
    if (n > cap) return perr(RK_ERR_LIMIT, "rk_bgzf_fastq_records: the records of the job need more bytes than the buffer holds");
    memcpy(dst, buf.data() + head, n);
    *nbytes = n;
    return RK_OK;
}

} // extern "C"
