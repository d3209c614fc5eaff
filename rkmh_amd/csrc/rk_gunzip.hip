// rk_gunzip.hip -- an ordinary gzip file (ONE deflate stream, not BGZF) inflated on the device, stretch after stretch.
//
// The reference opens every input with gzopen (/root/reference/src/rkmh.cpp:238-263): one sequential inflater, ~0.3 GB/s of text.
// A deflate stream can only be read from its beginning -- a match may reach 32 KB back and the code tables change block by block --
// but the two-pass inflater of rk_inflate.hip splits that dependency: pass 1 (a lane per chunk: Huffman decoding into a literal
// stream and match entries) needs no text at all, only a block header to start from; pass 2 (a wave per chunk: the text built from
// literals and matches) needs the 32 KB in front of the chunk -- and gets a stand-in pattern instead, three times over, from which
// the true bytes follow once the windows have been handed along the chunks in stream order (k_gz_windows: 32 KB per chunk, one
// workgroup).  What one call of gzip_next does with the next stretch of compressed bytes:
//   1. upload (DMA from the page-locked mapping of the file);
//   2. k_gz_find_starts: the first block header at or after every chunk boundary (RKMH_GZIP_CHUNK_KB, 32 KB of compressed bytes);
//   3. k_inflate_lanes<STREAM>: every chunk from its header to the first block boundary at or after the next chunk's;
//   4. the host follows the chain: a chunk belongs to the stream when the one before it ended exactly at its header (a false header
//      -- random bits that look like one -- is stepped over: the gap behind the chunk that met it is decoded in a launch of its own);
//   5. k_inflate_place<STREAM> x 3, k_gz_windows, k_gz_resolve: the text; k_crc32_segments: its CRC-32 in 64 KB segments, joined on
//      the host (the zero-advance matrices of rk_crc32.hpp) and compared with the file's trailer at the end of the stream, as is ISIZE;
//   6. the records: the text in front of the last record start (four-line rule) goes to the caller, the rest waits for the next call.
// Whatever the device cannot do -- a stored or fixed-code stretch longer than a chunk's scratch, text that outgrows the buffers, a
// second gzip member behind the first -- ends with return code 1 and the offset in the TEXT of the first record that was not
// delivered: the caller's sequential reader goes on from there (rk_reader_open_at).
#include "rk_api_internal.hpp"
#include "rk_kernels.hpp"
#include "rk_crc32.hpp"
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

using namespace rk;

namespace {
constexpr uint64_t CARRY_CAP = (uint64_t)4 << 20;  // text in front of a call's own: the records cut off the call before (a multiple of 16)
constexpr uint64_t OVER = (uint64_t)8 << 20;       // compressed bytes uploaded beyond a stretch: its last chunk ends at the first block boundary behind it
constexpr uint32_t TAIL_WINDOW = 1u << 18;         // the last record start is looked for this far in front of the text's end
constexpr uint32_t PATCH_MAX = 8;                  // gaps (behind false headers) decoded one launch each, per call
const Crc32Tables CRC_TABLES = make_crc32_tables();
}

struct rk_gzip {
    std::string path;
    int fd = -1;
    const uint8_t* map = nullptr;
    uint64_t size = 0, data_off = 0, data_end = 0; // deflate data = [data_off, data_end); the trailer follows
    uint32_t crc_want = 0, isize_want = 0;
    int first_byte = -1;
    // the plan (rk_gzip_plan): compressed bytes per call
    uint64_t stretch = 0;
    int64_t ncalls = 0;
    // where the stream stands
    int64_t next_call = 0;
    uint64_t next_bit = 0;     // in the file
    uint64_t text_made = 0;    // bytes of text inflated so far
    uint64_t text_given = 0;   // ... of which delivered (the rest is the carry)
    uint32_t carry_len = 0;
    uint32_t crc_run = 0;
    bool finished = false;     // the stream's last block has been decoded
    // device side (made at the first call, for the device of the slot that calls)
    rk_ctx* c = nullptr;
    DevBuf d_tmp; // what lives from call to call: [32 KB: the ring of the window behind the last chunk][its head][16: a cut][64 KB on: the carry]
};

extern "C" void rk_gzip_close(rk_gzip* gz) {
    if (!gz) return;
    if (gz->c) { hipError_t e = hipSetDevice(gz->c->device); (void)e; }
    gz->d_tmp.release();
    if (gz->map) munmap(const_cast<uint8_t*>(gz->map), (size_t)gz->size);
    if (gz->fd >= 0) close(gz->fd);
    delete gz;
}

// the device memory of a file that has been read (the window and the carry: 4 MB; the work buffers are the slot's); the next
// rk_gzip_plan + calls make it again
extern "C" void rk_gzip_release_device(rk_gzip* gz) {
    if (!gz) return;
    if (gz->c) { hipError_t e = hipSetDevice(gz->c->device); (void)e; }
    gz->d_tmp.release();
    gz->c = nullptr;
    gz->stretch = 0; // (a plan is needed before the next call)
}

// RK_OK: a regular file that holds a gzip member with a deflate payload (RFC 1952); the caller has ruled BGZF out before
extern "C" int rk_gzip_open(const char* path, rk_gzip** out) {
    if (!path || !out) return fail(RK_ERR_ARG, "bad arguments");
    *out = nullptr;
    rk_gzip* gz = new rk_gzip();
    struct Guard { rk_gzip* g; ~Guard() { if (g) rk_gzip_close(g); } } guard{gz};
    gz->path = path;
    gz->fd = open(path, O_RDONLY);
    struct stat st;
    if (gz->fd < 0 || fstat(gz->fd, &st) != 0 || !S_ISREG(st.st_mode)) return fail(RK_ERR_IO, "cannot open %s", path);
    gz->size = (uint64_t)st.st_size;
    if (gz->size < 18 + 2) return fail(RK_ERR_IO, "%s: not a gzip file", path);
    void* mp = mmap(nullptr, (size_t)gz->size, PROT_READ, MAP_SHARED, gz->fd, 0);
    if (mp == MAP_FAILED) return fail(RK_ERR_IO, "cannot map %s", path);
    gz->map = (const uint8_t*)mp;
    const uint8_t* p = gz->map;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return fail(RK_ERR_IO, "%s: not a gzip file", path);
    const uint8_t flg = p[3];
    uint64_t at = 10;
    if (flg & 4) { if (at + 2 > gz->size) return fail(RK_ERR_IO, "%s: truncated gzip header", path); at += 2 + ((uint64_t)p[at] | ((uint64_t)p[at + 1] << 8)); }
    for (int f : {8, 16}) // FNAME, FCOMMENT: zero-terminated
        if (flg & f) { while (at < gz->size && p[at]) ++at; ++at; }
    if (flg & 2) at += 2;
    if (at + 8 >= gz->size) return fail(RK_ERR_IO, "%s: truncated gzip header", path);
    gz->data_off = at; gz->data_end = gz->size - 8;
    const uint8_t* t = p + gz->data_end;
    gz->crc_want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    gz->isize_want = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    { // the text's first byte: zlib on the first few hundred compressed bytes
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) == Z_OK) {
            uint8_t o1[16];
            zs.next_in = const_cast<uint8_t*>(p + at); zs.avail_in = (uInt)std::min<uint64_t>(gz->data_end - at, 4096);
            zs.next_out = o1; zs.avail_out = 1;
            const int rc = inflate(&zs, Z_SYNC_FLUSH);
            if ((rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR) && zs.avail_out == 0) gz->first_byte = o1[0];
            inflateEnd(&zs);
        }
    }
    gz->next_bit = gz->data_off * 8;
    guard.g = nullptr;
    *out = gz;
    return RK_OK;
}
extern "C" const uint8_t* rk_gzip_image(const rk_gzip* gz) { return gz ? gz->map : nullptr; }
extern "C" uint64_t rk_gzip_file_bytes(const rk_gzip* gz) { return gz ? gz->size : 0; }
extern "C" int rk_gzip_first_byte(const rk_gzip* gz) { return gz ? gz->first_byte : -1; }
// the text's length as the trailer states it (ISIZE: modulo 2^32 -- a hint for sizing, nothing else, until the stream has been read)
extern "C" uint64_t rk_gzip_text_bytes_hint(const rk_gzip* gz) {
    if (!gz) return 0;
    uint64_t n = gz->isize_want;
    const uint64_t comp = gz->data_end - gz->data_off;
    while (n < comp) n += (uint64_t)1 << 32; // (text shorter than its deflate data is not what anybody compresses)
    return n;
}
// How many calls of rk_fastq_slot_load_gzip the file takes when the caller's slots hold slot_bytes of text: the compressed bytes are
// cut into equal stretches that should each inflate to ~85 % of a slot (by the file's own ratio).  Also rewinds the stream.
extern "C" int64_t rk_gzip_plan(rk_gzip* gz, uint64_t slot_bytes) {
    if (!gz || slot_bytes < ((uint64_t)1 << 20)) { fail(RK_ERR_ARG, "bad arguments (slots of at least 1 MB)"); return -1; }
    const uint64_t comp = gz->data_end - gz->data_off, text = rk_gzip_text_bytes_hint(gz);
    const double ratio = std::max(1.0, (double)text / (double)std::max<uint64_t>(comp, 1));
    uint64_t stretch = (uint64_t)((double)(slot_bytes - CARRY_CAP / 4) * 0.85 / ratio);
    stretch = std::min<uint64_t>(stretch, (uint64_t)400 << 20); // (bit offsets in a stretch are 32-bit)
    if (const char* e = getenv("RKMH_GZIP_STRETCH_KB")) { const long v = atol(e); if (v >= 16) stretch = (uint64_t)v << 10; } // (tests: many calls per file)
    stretch = std::max<uint64_t>(stretch, 16384) & ~(uint64_t)15;
    gz->stretch = stretch;
    gz->ncalls = (int64_t)((comp + stretch - 1) / stretch);
    if (gz->ncalls < 1) gz->ncalls = 1;
    gz->next_call = 0; gz->next_bit = gz->data_off * 8; gz->text_made = gz->text_given = 0; gz->carry_len = 0; gz->crc_run = 0; gz->finished = false;
    return gz->ncalls;
}
extern "C" int64_t rk_gzip_calls(const rk_gzip* gz) { return gz ? gz->ncalls : 0; }

// The slot's work buffers at the sizes a stretch of comp_bytes compressed bytes that inflates to at most cap_out bytes will ask for
// (rk_fastq_slot_reserve_gzip: a worker makes them when it starts -- one worker at a time -- as ONE arena: a runtime call of this kind
// takes 10 - 50 ms while other workers are setting up, whatever its size)
int gzip_reserve(GzScratch& S, rk_ctx* c, uint64_t comp_bytes, uint64_t cap_out) {
    RKCHK(set_dev(c));
    const uint64_t up = comp_bytes + OVER + 4096;
    const uint64_t chunks = up / 32768 + 16;
    struct Want { DevBuf* b; size_t bytes; };
    const Want dev[] = {
        {&S.d_stage, (size_t)CARRY_CAP + cap_out + 4096},
        {&S.d_comp, (size_t)up + 512},
        {&S.d_scratch, (size_t)(up + chunks * 8192 + (PATCH_MAX + 1) * ((uint64_t)3 << 20)) * 5 + (chunks + PATCH_MAX) * 4096 + 64},
        {&S.d_planes, (size_t)3 * (cap_out + 256) + 256},
        {&S.d_rings, (size_t)(std::min<uint64_t>(chunks, 3100) + 2) * 32768},
        {&S.d_heads, (size_t)(chunks + 8) * 4 + (size_t)((cap_out >> 16) + 8) * 4},
        {&S.d_chunks, (size_t)(2 * chunks + PATCH_MAX + 8) * sizeof(GzChunk)},
        {&S.d_misc, (size_t)(chunks + 1) * 12 + 64},
    };
    bool enough = true;
    size_t total = 0;
    for (const Want& w : dev) { if (w.b->cap < w.bytes) enough = false; total += (w.bytes + 4095) & ~(size_t)4095; }
    const size_t h_chunks = (size_t)(2 * chunks + PATCH_MAX + 8) * sizeof(GzChunk), h_misc = (16 + (size_t)(cap_out >> 16) + 8 + 3 * ((size_t)(comp_bytes >> 10) + 8)) * 4;
    if (enough && S.h_chunks.cap >= h_chunks && S.h_misc.cap >= h_misc) return RK_OK;
    for (const Want& w : dev) w.b->release();
    S.h_chunks.release(); S.h_misc.release();
    S.arena.release(); S.harena.release();
    RKCHK(S.arena.reserve(total + 4096));
    RKCHK(S.harena.reserve(((h_chunks + 4095) & ~(size_t)4095) + h_misc + 4096));
    size_t at = 0;
    for (const Want& w : dev) { w.b->set_view(S.arena.as<uint8_t>() + at, w.bytes); at += (w.bytes + 4095) & ~(size_t)4095; }
    S.h_chunks.set_view(S.harena.p, h_chunks);
    S.h_misc.set_view(S.harena.as<uint8_t>() + ((h_chunks + 4095) & ~(size_t)4095), h_misc);
    return RK_OK;
}
extern "C" uint64_t rk_gzip_stretch_bytes(const rk_gzip* gz) { return gz ? gz->stretch : 0; }

// The next stretch of gz's stream -> whole records at d_out (at most cap_out bytes), *nbytes of them; *text_off = the offset of the
// first of them in the file's text.  RK_OK; 1: not for the device from *text_off on (see the head of this file); < 0: an error
// (damaged data: a CRC-32 or length that does not match the trailer).  Calls come in order, call = 0 .. ncalls - 1, on stream st.
// raw: the stretch's TEXT, every byte of it, instead of its FASTQ records (references: rk_fasta_load_put_gzip) -- nothing is carried.
int gzip_next(rk_gzip* gz, GzScratch& S, rk_ctx* c, hipStream_t st, hipEvent_t ev, int64_t call, uint8_t* d_out, uint64_t cap_out, uint64_t* nbytes, uint64_t* text_off, bool raw) {
    static const bool timing = getenv("RKMH_BGZF_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    *nbytes = 0;
    *text_off = gz->text_given;
    if (gz->stretch == 0 || call != gz->next_call || call >= gz->ncalls) return fail(RK_ERR_STATE, "rk_fastq_slot_load_gzip: calls come in order, after rk_gzip_plan");
    if (gz->c && gz->c != c) { // another device than the pass before (-M: two passes over the file): the window and the carry move house
        if (call != 0) return fail(RK_ERR_STATE, "rk_fastq_slot_load_gzip: the calls of one pass come from slots of one device");
        RKCHK(set_dev(gz->c));
        gz->d_tmp.release();
    }
    gz->c = c;
    RKCHK(set_dev(c));
    ++gz->next_call;
    const bool last_call = call == gz->ncalls - 1;
    auto sync = [&]() -> int { HIPCHK(hipEventRecord(ev, st)); HIPCHK(hipEventSynchronize(ev)); return RK_OK; };
    double t_alloc = 0; // (time in allocations: zero from a slot's second file on)
#define GZ_RESERVE(buf, bytes) do { const auto t_a = std::chrono::steady_clock::now(); RKCHK((buf).reserve(bytes)); const double t_b = ms_since(t_a); t_alloc += t_b; \
        if (timing && t_b > 5.0) fprintf(stderr, "[gzip device] %s: %.0f MB reserved in %.1f ms\n", #buf, (double)(bytes) / 1e6, t_b); } while (0)
    // d_tmp lives from call to call and is made once, at its full size (a DevBuf that grows forgets what it held); the stage is the
    // slot's: [CARRY_CAP bytes: the carry is copied to their end][this call's text: up to cap_out bytes]
    GZ_RESERVE(gz->d_tmp, (size_t)65536 + CARRY_CAP + 64);
    GZ_RESERVE(S.d_stage, (size_t)CARRY_CAP + cap_out + 4096);
    // page-locked words of this call: [0, 16) cut and edge bytes, [16, 16 + segments) CRC-32 per 64 KB of text, then from / to / found per chunk boundary
    const size_t seg_max = (size_t)(cap_out >> 16) + 8, bound_max = (size_t)(gz->stretch >> 10) + 8;
    GZ_RESERVE(S.h_misc, (16 + seg_max + 3 * bound_max) * 4);
    uint32_t* const h_info = S.h_misc.as<uint32_t>();
    uint32_t* const h_crc = h_info + 16;
    // ---- the compressed bytes of this stretch
    const uint64_t range_end = std::min<uint64_t>(gz->data_off + (uint64_t)(call + 1) * gz->stretch, gz->data_end); // (file offset)
    const bool to_stream_end = last_call || range_end >= gz->data_end;
    uint32_t total = 0;
    std::vector<uint32_t> chain;
    uint32_t nchunk = 0;
    GzChunk* hc = nullptr;
    double t_find = 0, t_p1 = 0;
    if (!gz->finished && gz->next_bit < range_end * 8) {
        const uint64_t up0 = (gz->next_bit / 8) & ~(uint64_t)15;
        const uint64_t up1 = std::min<uint64_t>(gz->size, range_end + OVER);
        const uint64_t cbytes = up1 - up0;
        if (cbytes >= ((uint64_t)440 << 20)) return 1;
        const size_t cpad = ((size_t)cbytes + 15) & ~(size_t)15;
        GZ_RESERVE(S.d_comp, cpad + 256);
        uint8_t* const d_comp = S.d_comp.as<uint8_t>();
        HIPCHK(hipMemcpyAsync(d_comp, gz->map + up0, (size_t)cbytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(d_comp + cbytes, 0, cpad + 192 - cbytes, st));
        const uint32_t nbits = (uint32_t)((cpad + 64) * 8);
        const uint32_t first = (uint32_t)(gz->next_bit - up0 * 8), end_rel = (uint32_t)((range_end - up0) * 8);
        const uint32_t stop_last = to_stream_end ? 0xF0000000u : end_rel;
        // ---- chunk boundaries and the block headers behind them
        long ckb = 32;
        if (const char* e = getenv("RKMH_GZIP_CHUNK_KB")) { const long v = atol(e); if (v >= 1 && v <= 4096) ckb = v; }
        const uint32_t chunk_bits = (uint32_t)ckb << 13;
        const uint32_t nbound = end_rel > first ? (end_rel - first - 1) / chunk_bits : 0; // boundaries first + k chunk_bits < end_rel, k >= 1
        const size_t cap_chunks = 2 * ((size_t)nbound + 1) + PATCH_MAX + 8; // (the chunks, and behind them the units pass 2 groups them into)
        if ((size_t)nbound + 1 > bound_max) return fail(RK_ERR_STATE, "rk_fastq_slot_load_gzip: more chunk boundaries than planned");
        GZ_RESERVE(S.d_misc, (size_t)(nbound + 1) * 12 + 64);
        uint32_t* const h_from = h_crc + seg_max;
        uint32_t* const h_to = h_from + nbound + 1;
        uint32_t* const h_found = h_to + nbound + 1;
        uint32_t* const d_from = S.d_misc.as<uint32_t>();
        uint32_t* const d_to = d_from + nbound + 1;
        uint32_t* const d_found = d_to + nbound + 1;
        for (uint32_t k = 0; k < nbound; ++k) { h_from[k] = first + (k + 1) * chunk_bits; h_to[k] = std::min<uint32_t>(end_rel, first + (k + 2) * chunk_bits); }
        if (nbound) {
            HIPCHK(hipMemcpyAsync(d_from, h_from, (size_t)(nbound + 1) * 8, hipMemcpyHostToDevice, st));
            HIPCHK(launch_gz_find_starts(d_comp, nbits, d_from, d_to, nbound, d_found, st));
            HIPCHK(hipMemcpyAsync(h_found, d_found, (size_t)nbound * 4, hipMemcpyDeviceToHost, st));
        }
        RKCHK(sync());
        t_find = ms_since(t_0);
        // ---- pass 1
        GZ_RESERVE(S.h_chunks, cap_chunks * sizeof(GzChunk));
        GZ_RESERVE(S.d_chunks, cap_chunks * sizeof(GzChunk));
        hc = S.h_chunks.as<GzChunk>();
        std::vector<uint32_t> starts;
        starts.push_back(first);
        for (uint32_t k = 0; k < nbound; ++k) if (h_found[k] != 0xFFFFFFFFu && h_found[k] > starts.back()) starts.push_back(h_found[k]);
        nchunk = (uint32_t)starts.size();
        // a chunk's scratch: 5 bytes per compressed byte (a literal of 4 bits: 2 bytes per byte; a match of 10 bits: 3.2) and a little
        // for what it writes last.  A chunk normally stops exactly at its stop position -- the next chunk's header; one that runs on (a
        // false header in its way; the stretch's last chunk) or is denser than that overflows (status 22) and is decoded again, alone,
        // into a roomy region.  (The first form gave every chunk 128 KB of slack: 5.6 GB for a file of 183 MB -- and allocations of
        // that size take 0.15 - 0.8 s while other workers set up.)
        uint64_t sdw = 0;
        auto region_of = [](uint64_t bits, uint64_t slack) { return (uint32_t)(((bits / 8 + slack) * 5 + 4096) / 4); };
        for (uint32_t i = 0; i < nchunk; ++i) {
            GzChunk& g = hc[i];
            memset(&g, 0, sizeof g);
            g.start_bit = starts[i]; g.stop_bit = i + 1 < nchunk ? starts[i + 1] : stop_last;
            const uint32_t span = (i + 1 < nchunk ? starts[i + 1] : end_rel) - starts[i];
            g.scratch_off = (uint32_t)sdw; g.scratch_dw = region_of(span, i + 1 < nchunk ? 8192 : 262144);
            g.flags = (i == 0 && gz->text_made == 0) ? 1u : 0u;
            sdw += g.scratch_dw;
        }
        const uint64_t patch_dw = region_of(2 * (uint64_t)chunk_bits + OVER * 8 / 4, 262144);
        if (sdw + patch_dw * PATCH_MAX >= ((uint64_t)1 << 32)) return 1;
        GZ_RESERVE(S.d_scratch, (size_t)(sdw + patch_dw * PATCH_MAX) * 4 + 64);
        GzChunk* const d_chunks = S.d_chunks.as<GzChunk>();
        HIPCHK(hipMemcpyAsync(d_chunks, hc, (size_t)nchunk * sizeof(GzChunk), hipMemcpyHostToDevice, st));
        HIPCHK(launch_gz_lanes(d_comp, (uint32_t)(cpad + 64), d_chunks, nchunk, S.d_scratch.as<uint32_t>(), st));
        HIPCHK(hipMemcpyAsync(hc, d_chunks, (size_t)nchunk * sizeof(GzChunk), hipMemcpyDeviceToHost, st));
        RKCHK(sync());
        // ---- the chain: from the first chunk (a true header: the stream's first, or where the call before ended) on
        uint32_t cur = 0, patches = 0;
        for (;;) {
            const GzChunk& g = hc[cur];
            if (g.status == 22u && patches < PATCH_MAX && g.scratch_dw < (uint32_t)patch_dw) { // its region was too small: once more, alone, into a roomy one
                GzChunk& p = hc[nchunk];
                p = g;
                p.scratch_off = (uint32_t)(sdw + patch_dw * patches); p.scratch_dw = (uint32_t)patch_dw;
                p.status = 0;
                if (timing) fprintf(stderr, "[gzip device] the chunk at bit %u outgrew its scratch region: decoded again on its own\n", p.start_bit);
                HIPCHK(hipMemcpyAsync(d_chunks + nchunk, &p, sizeof(GzChunk), hipMemcpyHostToDevice, st));
                HIPCHK(launch_gz_lanes(d_comp, (uint32_t)(cpad + 64), d_chunks + nchunk, 1, S.d_scratch.as<uint32_t>(), st));
                HIPCHK(hipMemcpyAsync(&p, d_chunks + nchunk, sizeof(GzChunk), hipMemcpyDeviceToHost, st));
                RKCHK(sync());
                cur = nchunk++;
                ++patches;
                continue;
            }
            if (g.status != 0) { if (timing) fprintf(stderr, "[gzip device] chunk at bit %u: status %u\n", g.start_bit, g.status); return 1; }
            chain.push_back(cur);
            if ((uint64_t)total + g.out_len >= ((uint64_t)1 << 31)) return 1;
            total += g.out_len;
            if (g.final_seen) { gz->finished = true; gz->next_bit = up0 * 8 + g.end_bit; break; }
            const uint32_t e = g.end_bit;
            if (e >= end_rel) { gz->next_bit = up0 * 8 + e; break; } // (to_stream_end: only a final block ends the chain)
            const auto it = std::lower_bound(starts.begin(), starts.end(), e);
            if (it != starts.end() && *it == e) { cur = (uint32_t)(it - starts.begin()); continue; }
            // the chunk ran over a header that was none: the stretch from its end to the next header, in a launch of its own
            if (patches == PATCH_MAX) return 1;
            GzChunk& p = hc[nchunk];
            memset(&p, 0, sizeof p);
            p.start_bit = e; p.stop_bit = it != starts.end() ? *it : stop_last;
            p.scratch_off = (uint32_t)(sdw + patch_dw * patches); p.scratch_dw = (uint32_t)patch_dw;
            if (timing) fprintf(stderr, "[gzip device] a false block header before bit %u: the gap %u .. %u is decoded on its own\n", e, e, p.stop_bit);
            HIPCHK(hipMemcpyAsync(d_chunks + nchunk, &p, sizeof(GzChunk), hipMemcpyHostToDevice, st));
            HIPCHK(launch_gz_lanes(d_comp, (uint32_t)(cpad + 64), d_chunks + nchunk, 1, S.d_scratch.as<uint32_t>(), st));
            HIPCHK(hipMemcpyAsync(&p, d_chunks + nchunk, sizeof(GzChunk), hipMemcpyDeviceToHost, st));
            RKCHK(sync());
            cur = nchunk++;
            ++patches;
        }
        t_p1 = ms_since(t_0);
    }
    // ---- pass 2: the chain's chunks side by side
    const uint32_t carry = gz->carry_len;
    if ((uint64_t)carry + total > cap_out + CARRY_CAP / 2 || total > cap_out) return 1; // (more text than the file's ratio promised)
    const size_t plane_stride = ((size_t)total + 15 + 64) & ~(size_t)15;
    uint8_t* const text = S.d_stage.as<uint8_t>() + CARRY_CAP; // this call's text; the carry ends where it begins
    if (carry) HIPCHK(hipMemcpyAsync(text - carry, gz->d_tmp.as<uint8_t>() + 65536, carry, hipMemcpyDeviceToDevice, st));
    const uint32_t nch = (uint32_t)chain.size();
    const uint32_t nseg = (total + 65535u) >> 16;
    if (nseg > seg_max) return 1;
    if (nch) {
        GZ_RESERVE(S.d_planes, 3 * plane_stride + 256);
        const uint32_t group = std::min<uint32_t>(8u, std::max<uint32_t>(1u, (nch + 3071u) / 3072u)); // (chunks per unit of pass 2: see below)
        const uint32_t nunits = (nch + group - 1) / group;
        GZ_RESERVE(S.d_rings, (size_t)(nunits + 1) * 32768);
        GZ_RESERVE(S.d_heads, (size_t)(nch + 2) * 4 + (size_t)(nseg + 1) * 4);
        // (rings[0] / heads[0]: the window the call before left -- see the end of this function; the stream's first call: zeros)
        if (gz->text_made == 0) { HIPCHK(hipMemsetAsync(gz->d_tmp.p, 0, 32768 + 4, st)); }
        HIPCHK(hipMemcpyAsync(S.d_rings.p, gz->d_tmp.p, 32768, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(S.d_heads.p, gz->d_tmp.as<uint8_t>() + 32768, 4, hipMemcpyDeviceToDevice, st));
        // the chain's chunks in order with their places, then the UNITS pass 2 works in: `group` consecutive chunks placed as one
        // stretch of text (a step of the sequential window walk per unit: ~3 000 units keep the CUs busy and the walk short)
        std::vector<GzChunk> cc((size_t)nch + nunits);
        uint32_t off = 0;
        for (uint32_t i = 0; i < nch; ++i) { cc[i] = hc[chain[i]]; cc[i].out_off = off; off += cc[i].out_len; }
        for (uint32_t uix = 0; uix < nunits; ++uix) {
            GzChunk& un = cc[(size_t)nch + uix];
            memset(&un, 0, sizeof un);
            un.out_off = cc[(size_t)uix * group].out_off;
            for (uint32_t j = uix * group; j < std::min(nch, (uix + 1) * group); ++j) un.out_len += cc[j].out_len;
        }
        GZ_RESERVE(S.h_chunks, cc.size() * sizeof(GzChunk));
        GZ_RESERVE(S.d_chunks, cc.size() * sizeof(GzChunk));
        hc = S.h_chunks.as<GzChunk>();
        memcpy(hc, cc.data(), cc.size() * sizeof(GzChunk)); // (page-locked: the upload reads it after this function's next lines)
        GzChunk* const d_chunks = S.d_chunks.as<GzChunk>();
        HIPCHK(hipMemcpyAsync(d_chunks, hc, cc.size() * sizeof(GzChunk), hipMemcpyHostToDevice, st));
        uint32_t* const d_crc = S.d_heads.as<uint32_t>() + nunits + 2;
        HIPCHK(launch_gz_place(d_chunks, nch, group, d_chunks + nch, nunits, S.d_planes.as<uint8_t>(), plane_stride, S.d_scratch.as<uint32_t>(), S.d_rings.as<uint8_t>(),
                               S.d_heads.as<uint32_t>(), text, total, d_crc, st));
        // the window behind this call's last chunk, for the next call (d_tmp: 32 KB of ring + its head)
        HIPCHK(hipMemcpyAsync(gz->d_tmp.p, S.d_rings.as<uint8_t>() + (size_t)nunits * 32768, 32768, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(gz->d_tmp.as<uint8_t>() + 32768, S.d_heads.as<uint32_t>() + nunits, 4, hipMemcpyDeviceToDevice, st));
        if (nseg) HIPCHK(hipMemcpyAsync(h_crc, d_crc, (size_t)nseg * 4, hipMemcpyDeviceToHost, st));
    }
    // ---- the records: everything in front of the last record start near the end (all of it at the end of the stream)
    const uint32_t ntext = carry + total;
    const uint8_t* const all = text - carry;
    const bool at_end = gz->finished || (call == gz->ncalls - 1);
    uint32_t* const d_cuts = gz->d_tmp.as<uint32_t>() + (32768 + 16) / 4;
    h_info[0] = ntext; h_info[1] = 0;
    if (!at_end && ntext && !raw) {
        const uint32_t from = ntext > TAIL_WINDOW ? ntext - TAIL_WINDOW : 0;
        HIPCHK(launch_fastq_first_start(all, ntext, from, TAIL_WINDOW, false, d_cuts, 0, st));
        HIPCHK(hipMemcpyAsync(h_info, d_cuts, 4, hipMemcpyDeviceToHost, st));
    }
    if (ntext) {
        HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(h_info + 1), all, 1, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(h_info + 1) + 1, all + ntext - 1, 1, hipMemcpyDeviceToHost, st));
    }
    RKCHK(sync());
    if (const char* dump = getenv("RKMH_GZIP_DUMP")) { // (debugging: this call's text, as the device built it, appended to a file)
        std::vector<uint8_t> h(total);
        if (total) HIPCHK(hipMemcpy(h.data(), text, total, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dump, "ab")) { fwrite(h.data(), 1, h.size(), f); fclose(f); }
        if (timing) for (uint32_t i = 0; i < nch; ++i) fprintf(stderr, "[gzip device] chunk %u: bits %u .. %u, text %u + %u, %u entries, %u literals\n", i, hc[i].start_bit, hc[i].end_bit, hc[i].out_off, hc[i].out_len, hc[i].nent, hc[i].nlit);
    }
    // the stream's CRC-32 so far; at its end, the trailer's word on it
    for (uint32_t sg = 0; sg < nseg; ++sg) {
        const uint32_t len = std::min<uint32_t>(65536u, total - (sg << 16));
        // CRC(A B) = advance(CRC(A), |B| zero bytes) ^ CRC(B) -- what zlib's crc32_combine computes; zlib 1.2.11 rebuilds its operator
        // matrices in every call (~10 us: 0.1 s for the 11 000 segments of a 700 MB stretch), the tables of rk_crc32.hpp hold them
        gz->crc_run = crc32_advance(CRC_TABLES, gz->crc_run, len) ^ h_crc[sg];
        gz->text_made += len;
    }
    if (gz->finished) {
        const uint64_t end_byte = (gz->next_bit + 7) / 8;
        if (end_byte != gz->data_end) { // more behind the stream's end: another member (the sequential reader handles those), or debris
            if (timing) fprintf(stderr, "[gzip device] %s: the deflate stream ends at byte %llu, the file's trailer begins at %llu\n", gz->path.c_str(), (unsigned long long)end_byte, (unsigned long long)gz->data_end);
            return 1;
        }
        if (gz->crc_run != gz->crc_want || (uint32_t)gz->text_made != gz->isize_want)
            return fail(RK_ERR_IO, "%s: the inflated text does not match the file's trailer (CRC-32 %08x, stated %08x; length %llu, stated %u modulo 2^32): damaged data",
                        gz->path.c_str(), gz->crc_run, gz->crc_want, (unsigned long long)gz->text_made, gz->isize_want);
    } else if (call == gz->ncalls - 1) return fail(RK_ERR_IO, "%s: the deflate stream does not end inside the file: truncated?", gz->path.c_str());
    const uint8_t first_byte = reinterpret_cast<const uint8_t*>(h_info + 1)[0], last_byte = reinterpret_cast<const uint8_t*>(h_info + 1)[1];
    uint32_t cut = (at_end || raw) ? ntext : h_info[0];
    if (cut == 0xFFFFFFFFu) return 1;          // no record start in the tail window: records longer than 256 KB, or no FASTQ
    if (cut > ntext) cut = ntext;
    if (!raw && ntext && gz->text_given == 0 && first_byte != '@') return 1;
    if (cut > cap_out - (raw ? 0 : 1) || ntext - cut > CARRY_CAP) return 1;
    if (cut) HIPCHK(hipMemcpyAsync(d_out, all, cut, hipMemcpyDeviceToDevice, st));
    uint64_t n = cut;
    if (!raw && at_end && ntext && last_byte != '\n') { HIPCHK(hipMemsetAsync(d_out + n, '\n', 1, st)); ++n; } // a last line without its newline
    // the carry: the text behind the cut waits in the file's own buffer for the next call
    const uint32_t new_carry = ntext - cut;
    if (new_carry) HIPCHK(hipMemcpyAsync(gz->d_tmp.as<uint8_t>() + 65536, all + cut, new_carry, hipMemcpyDeviceToDevice, st));
    gz->carry_len = new_carry;
    *text_off = gz->text_given;
    gz->text_given += cut;
    *nbytes = n;
    if (timing) fprintf(stderr, "[gzip device] %s call %lld of %lld: %u chunks (%u in the chain), %.1f MB of text; headers %.1f ms, pass 1 %.1f, all %.1f (allocations %.1f)\n", gz->path.c_str(),
                        (long long)call + 1, (long long)gz->ncalls, nchunk, nch, total / 1e6, t_find, t_p1 - t_find, ms_since(t_0), t_alloc);
#undef GZ_RESERVE
    return RK_OK;
}
