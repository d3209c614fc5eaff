// rk_frontend.hip -- the read and reference front ends ON THE DEVICE behind the C ABI (include/rkmh_amd.h):
//   * FASTQ slots: a block of raw FASTQ text is split into records, checked, packed and classified by the GPU (rk_fastq.hip) -- the
//     device-side replacement of parse_fastas -> kseq_read (/root/reference/src/rkmh.cpp:238-263) + the per-read loop (:845-898);
//   * BGZF jobs inflated on the device (rk_inflate.hip): the compressed members cross the link, the text is built, CRC-checked and
//     cut to records in HBM, and -- slots created with RK_SLOT_DEVICE_TEXT -- never visits the host: what comes back per read is
//     its row, and the bytes the output needs (the names for stream's lines, the passing records for filter), packed on the device;
//   * reference FASTA text stripped on the device (rk_fasta.hip).
#include "rk_api_internal.hpp"

// a host array that a device copy lands in: page-locked up to the size reserved at creation, pageable beyond (a block with far more
// records than blocks of its size usually hold must not fail -- and must not make every slot page-lock memory for the worst case)
struct HostArr {
    PinBuf pin;
    std::vector<uint8_t> big;
    int reserve_pinned(size_t bytes) { return pin.reserve(bytes); }
    void* get(size_t bytes) {
        if (bytes <= pin.cap) return pin.p;
        if (big.size() < bytes) big.resize(bytes + bytes / 8);
        return big.data();
    }
    void release() { pin.release(); std::vector<uint8_t>().swap(big); }
};

// ------------------------------------------------------------------------------------------------
// FASTQ text parsed on the device (rk_fastq.hip): one slot = one block in flight (its own stream, host buffers, device arrays).
// Several slots of one context may be driven from several host threads at once.
struct rk_fastq_slot {
    rk_ctx* c = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr; // blocking-sync event: a dozen host threads wait for their blocks at once and must SLEEP while they
                             // do (hipStreamSynchronize spins: the waiting threads would take the cores the reading and formatting ones need)
    uint64_t max_bytes = 0;
    int flags = 0;
    bool device_text() const { return (flags & RK_SLOT_DEVICE_TEXT) != 0; }
    PinBuf h_text, h_info, h_mem;
    HostArr h_out4, h_spans, h_pack;
    DevBuf d_text, d_u32, d_bases, d_out4, d_scan, d_pack;
    FqDev d{};
    uint64_t pending = 0;   // bytes of the block between submit and finish
    bool submitted = false;
    // BGZF members inflated on the device (rk_fastq_slot_load_bgzf): compressed bytes + member table up, text built in d_inf, the
    // job's records moved to d_text -- the next submit / count then skips its upload (text_on_device)
    DevBuf d_comp, d_mem, d_inf, d_match;
    bool text_on_device = false;
    // rk_fastq_slot_set_source: the block's text lies in caller memory (a page-locked mapping of the file): the next submit uploads it
    // from there, and the formatters are given that pointer
    const uint8_t* src = nullptr;      // of the block in flight (nullptr: h_text)
    const uint8_t* next_src = nullptr; // armed for the next submit
    // RK_SLOT_DEVICE_TEXT: what is packed for the host -- the names (stream / classify), or the records filter prints
    bool pack_filter = false;
    int min_matches = -1, min_diff = 0;
    const uint8_t* spans_base = nullptr; // what the spans of the last finished block index (rk_fastq_slot_spans_base)
    GzScratch* gzs = nullptr;            // the work buffers of rk_fastq_slot_load_gzip (made at its first call)
};

extern "C" void rk_fastq_slot_destroy(rk_fastq_slot* s) {
    if (!s) return;
    if (s->c) { hipError_t e = hipSetDevice(s->c->device); (void)e; }
    if (s->st) { hipError_t e = hipStreamSynchronize(s->st); (void)e; e = hipStreamDestroy(s->st); (void)e; }
    if (s->ev) { hipError_t e = hipEventDestroy(s->ev); (void)e; }
    for (PinBuf* b : {&s->h_text, &s->h_info, &s->h_mem}) b->release();
    for (HostArr* b : {&s->h_out4, &s->h_spans, &s->h_pack}) b->release();
    for (DevBuf* b : {&s->d_text, &s->d_u32, &s->d_bases, &s->d_out4, &s->d_scan, &s->d_pack, &s->d_comp, &s->d_mem, &s->d_inf, &s->d_match}) b->release();
    if (s->gzs) { s->gzs->release(); delete s->gzs; }
    delete s;
}

extern "C" int rk_fastq_slot_create2(rk_ctx* c, uint64_t max_bytes, int flags, rk_fastq_slot** out) {
    if (!c || !out || max_bytes < 4096 || max_bytes > ((uint64_t)1 << 31)) return fail(RK_ERR_ARG, "bad arguments (block size 4 KB .. 2 GB)");
    if (flags & ~RK_SLOT_DEVICE_TEXT) return fail(RK_ERR_ARG, "unknown slot flags %d", flags);
    RKCHK(set_dev(c));
    rk_fastq_slot* s = new rk_fastq_slot();
    s->c = c; s->max_bytes = max_bytes; s->flags = flags;
    struct Guard { rk_fastq_slot* s; ~Guard() { if (s) rk_fastq_slot_destroy(s); } } guard{s};
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->ev, hipEventBlockingSync | hipEventDisableTiming));
    // capacities: records of fewer than 64 bytes on average (reads of about 25 bases) make the block "irregular" (FQ_BAD_CAP) --
    // the host scanner takes it -- instead of sizing every device array for the worst case.  The page-locked host arrays of a
    // device-text slot (hundreds of megabytes of text per block; page-locking costs ~0.2 ms per MB at start-up) are sized for
    // records of 256 bytes and names of a twentieth of the text; a block beyond that lands in pageable memory (HostArr).
    const uint32_t chunks = (uint32_t)((max_bytes + 4095) / 4096);
    const uint32_t rec_cap = (uint32_t)(max_bytes / 64 + 64), line_cap = 4 * rec_cap + 16;
    const bool dt = s->device_text();
    static const bool timing = getenv("RKMH_BGZF_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const size_t host_recs = dt ? (size_t)(max_bytes / 256 + 4096) : (size_t)rec_cap;
    if (!dt) RKCHK(s->h_text.reserve(max_bytes + 64));
    RKCHK(s->h_out4.reserve_pinned(host_recs * 16));
    RKCHK(s->h_spans.reserve_pinned(host_recs * (dt ? 8 : 20)));
    RKCHK(s->h_info.reserve(64));
    if (dt) RKCHK(s->h_pack.reserve_pinned(std::max<uint64_t>((uint64_t)1 << 20, max_bytes / 20) + 64));
    const double t_pinned = ms_since(t_0);
    RKCHK(s->d_text.reserve(max_bytes + 64));
    RKCHK(s->d_bases.reserve(max_bytes + 64));
    RKCHK(s->d_out4.reserve((size_t)rec_cap * 16));
    const size_t n32 = (size_t)2 * (chunks + 1) + line_cap + (size_t)(dt ? 8 : 6) * (rec_cap + 1) + 4;
    RKCHK(s->d_u32.reserve(n32 * 4));
    const size_t tb = fq_scan_temp_bytes(std::max(chunks + 1, rec_cap + 1));
    RKCHK(s->d_scan.reserve(tb));
    uint32_t* u = s->d_u32.as<uint32_t>();
    FqDev& d = s->d;
    d.chunk_cnt = u; u += chunks + 1;
    d.chunk_base = u; u += chunks + 1;
    d.nl = u; u += line_cap;
    d.seq_off = u; u += rec_cap + 1;
    d.seq_len = u; u += rec_cap + 1;
    d.qual_off = u; u += rec_cap + 1;
    d.name_off = u; u += rec_cap + 1;
    d.name_len = u; u += rec_cap + 1;
    d.out_off = u; u += rec_cap + 1;
    if (dt) { d.pack_len = u; u += rec_cap + 1; d.pack_off = u; u += rec_cap + 1; }
    d.info = u;
    d.line_cap = line_cap; d.rec_cap = rec_cap;
    d.bases = s->d_bases.as<uint8_t>();
    d.scan_tmp = s->d_scan.p; d.scan_tmp_bytes = tb;
    if (dt) {
        // (names: an eighth of the text at most -- a block with more is left to the host scanner; the records filter prints may be all of it: rk_fastq_slot_set_filter_output)
        RKCHK(s->d_pack.reserve(max_bytes / 8 + 64));
        d.pack = s->d_pack.as<uint8_t>(); d.pack_cap = max_bytes / 8;
        // the inflater's buffers, at the sizes rk_fastq_slot_load_bgzf asks for: made here (beside the caller's reference stage),
        // not in front of the slot's first job
        const uint64_t cap_text = max_bytes + 5 * 65536ull + 64, cap_mem = cap_text / 16384 + 16;
        const size_t mem_bytes = (((size_t)cap_mem * sizeof(InflateMember) + 15) & ~(size_t)15) + (size_t)cap_mem * 8 + 64;
        RKCHK(s->h_mem.reserve(mem_bytes));
        RKCHK(s->d_mem.reserve(mem_bytes));
        RKCHK(s->d_comp.reserve(cap_text * 5 / 8 + 256));
        RKCHK(s->d_inf.reserve(cap_text));
        RKCHK(s->d_match.reserve(cap_text * 4 / 3 + cap_mem * 32 + 64));
    }
    const double t_device = ms_since(t_0) - t_pinned;
    HIPCHK(hipMemsetAsync(s->d_u32.p, 0, n32 * 4, s->st)); // stale lengths past a block's last record must at least be defined
    HIPCHK(hipStreamSynchronize(s->st));
    if (timing && dt) fprintf(stderr, "[bgzf device] slot of %.0f MB: page-locked arrays %.1f ms, device arrays %.1f ms, clear %.1f ms\n", (double)max_bytes / 1e6, t_pinned, t_device,
                              ms_since(t_0) - t_pinned - t_device);
    guard.s = nullptr;
    *out = s;
    return RK_OK;
}
extern "C" int rk_fastq_slot_create(rk_ctx* c, uint64_t max_bytes, rk_fastq_slot** out) { return rk_fastq_slot_create2(c, max_bytes, 0, out); }

extern "C" uint8_t* rk_fastq_slot_text(rk_fastq_slot* s) { return s ? s->h_text.as<uint8_t>() : nullptr; }
extern "C" const uint8_t* rk_fastq_slot_spans_base(const rk_fastq_slot* s) { return s ? s->spans_base : nullptr; }
extern "C" int rk_fastq_slot_set_filter_output(rk_fastq_slot* s, int min_matches, int min_diff) {
    if (!s || !s->device_text()) return fail(RK_ERR_ARG, "rk_fastq_slot_set_filter_output: a slot created with RK_SLOT_DEVICE_TEXT is needed");
    if (s->d.pack_cap < s->max_bytes) {
        RKCHK(set_dev(s->c));
        HIPCHK(hipStreamSynchronize(s->st));
        RKCHK(s->d_pack.reserve(s->max_bytes + 64));
        s->d.pack = s->d_pack.as<uint8_t>(); s->d.pack_cap = s->max_bytes;
    }
    s->pack_filter = true; s->min_matches = min_matches; s->min_diff = min_diff;
    return RK_OK;
}
// The NEXT block of this slot is read from `text` (caller memory that stays valid and unchanged until the block's finish / count
// has returned) instead of the slot's own buffer: a page-locked mapping of the input file (mmap + hipHostRegister) lets the DMA
// engine read the page cache itself -- no pread copy (tools/ubench/mmap_register.hip: 55 GB/s against 18-20 for one thread's pread + upload).
extern "C" int rk_fastq_slot_set_source(rk_fastq_slot* s, const uint8_t* text) {
    if (!s) return fail(RK_ERR_ARG, "slot is NULL");
    s->next_src = text;
    return RK_OK;
}

// A BGZF job inflated ON THE DEVICE (rk_inflate.hip): the compressed bytes of members [lead(b0), b1 + 2) go up -- 0.58 x the text for
// level-1 FASTQ; straight from the mapped file when the caller page-locked it (rk_host_register_readonly on rk_bgzf_image) --, a lane
// per member decodes, a wave per member places the text and another checks its CRC-32 against the member's footer, the first
// record starts at or after the text of b0 and of b1 are found by the four-line rule (k_fastq_first_start: the rule of
// rk_bgzf_fastq_records, so host-inflated and device-inflated jobs agree), and the records between them are moved to the slot's
// device text buffer.  A slot WITHOUT RK_SLOT_DEVICE_TEXT also gets a copy in rk_fastq_slot_text().  The NEXT rk_fastq_slot_submit
// / _classify / _count of this slot takes *nbytes and skips its upload.
// Returns RK_OK, or 1: this job is for the host route (rk_bgzf_fastq_records) -- a member the device could not inflate or whose
// CRC-32 does not match (the host inflater then reports the damage), text that does not begin with '@', a record that outgrows
// the lookahead or the slot.
extern "C" int rk_fastq_slot_load_bgzf(rk_fastq_slot* s, const rk_bgzf* z, int64_t b0, int64_t b1, uint64_t* nbytes, uint64_t* text_off) {
    static const bool timing = getenv("RKMH_BGZF_TIMING") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    if (!nbytes) return fail(RK_ERR_ARG, "bad arguments");
    *nbytes = 0;
    if (!s || !z || b0 < 0 || b1 <= b0 || b1 > rk_bgzf_members(z)) return fail(RK_ERR_ARG, "bad arguments");
    if (text_off) *text_off = rk_bgzf_text_offset(z, b0);
    s->text_on_device = false;
    rk_ctx* c = s->c;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    const int64_t nb = rk_bgzf_members(z);
    const int64_t lo = rk_bgzf_lead_member(z, b0), ext = std::min<int64_t>(nb, b1 + 2);
    const uint32_t nm = (uint32_t)(ext - lo);
    uint64_t f_lo = 0, f_hi = 0;
    uint32_t tot = 0, hd = 0, us = 0;
    RKCHK(rk_bgzf_member(z, lo, &f_lo, &tot, &hd, &us));
    RKCHK(rk_bgzf_member(z, ext - 1, &f_hi, &tot, &hd, &us));
    const uint64_t cbytes = f_hi + tot - f_lo;
    const uint64_t u_lo = rk_bgzf_text_offset(z, lo), u_b0 = rk_bgzf_text_offset(z, b0), u_b1 = rk_bgzf_text_offset(z, b1), u_ext = rk_bgzf_text_offset(z, ext);
    const uint64_t ntext = u_ext - u_lo;
    if (ntext > s->max_bytes + 4 * 65536ull || cbytes >= ((uint64_t)1 << 31) || ntext >= ((uint64_t)1 << 31)) return 1;
    // (sized for the slot, not for this job: every job of a file is a little different, and growing a buffer job by job costs more than the job)
    const uint64_t cap_text = s->max_bytes + 5 * 65536ull + 64, cap_mem = std::max<uint64_t>(nm, cap_text / 16384 + 16);
    const size_t mem_bytes = (((size_t)cap_mem * sizeof(InflateMember) + 15) & ~(size_t)15) + (size_t)cap_mem * 8 + 64;
    RKCHK(s->h_mem.reserve(mem_bytes));
    RKCHK(s->d_mem.reserve(mem_bytes));
    RKCHK(s->d_comp.reserve(std::max<uint64_t>(cbytes, cap_text * 5 / 8) + 256));
    RKCHK(s->d_inf.reserve(cap_text));
    InflateMember* mt = s->h_mem.as<InflateMember>();
    uint64_t scratch_dw = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        uint64_t fo = 0;
        RKCHK(rk_bgzf_member(z, lo + i, &fo, &tot, &hd, &us));
        if (tot < hd + 8u) return fail(RK_ERR_IO, "BGZF member %lld is shorter than its header and footer", (long long)(lo + i));
        mt[i].in_off = (uint32_t)(fo - f_lo) + hd; mt[i].in_len = tot - hd - 8;
        mt[i].out_off = (uint32_t)(rk_bgzf_text_offset(z, lo + i) - u_lo); mt[i].out_len = us;
        mt[i].match_off = (uint32_t)scratch_dw; mt[i].pad = 0;
        scratch_dw += inflate_scratch_dwords(us);
    }
    if (scratch_dw >= ((uint64_t)1 << 32)) return 1;
    RKCHK(s->d_match.reserve(std::max<uint64_t>(scratch_dw * 4 + 64, cap_text * 4 / 3 + cap_mem * 32 + 64)));
    const double t_reserve = ms_since(t_0);
    // the compressed bytes: by DMA from where the file is mapped when the caller page-locked the mapping, else through the
    // runtime's own staging of pageable memory (no page-locked copy of ours: that buffer would be as large as the job)
    const uint8_t* image = rk_bgzf_image(z) + f_lo;
    HIPCHK(hipMemcpyAsync(s->d_comp.p, image, cbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(s->d_comp.as<uint8_t>() + cbytes, 0, 160, st)); // (the lanes of pass 1 request whole 16-byte pieces a little past their member)
    const size_t cpad = ((cbytes + 15) & ~(size_t)15) + 64;
    HIPCHK(hipMemcpyAsync(s->d_mem.p, mt, (size_t)nm * sizeof(InflateMember), hipMemcpyHostToDevice, st));
    const size_t status_at = ((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15;
    uint32_t* d_status = reinterpret_cast<uint32_t*>(s->d_mem.as<uint8_t>() + status_at);
    uint32_t* h_status = reinterpret_cast<uint32_t*>(s->h_mem.as<uint8_t>() + status_at);
    HIPCHK(launch_inflate_members(s->d_comp.as<uint8_t>(), (uint32_t)cpad, s->d_mem.as<InflateMember>(), nm, s->d_inf.as<uint8_t>(), s->d_match.as<uint32_t>(), d_status, st));
    // the cuts: cuts[0] = head, cuts[1] = tail (in the inflated text of members lo .. ext)
    uint32_t* d_cuts = s->d.info; // (the index kernels write it afterwards)
    const bool at_eof = ext == nb;
    const bool cut_head = u_b0 > 0, cut_tail = b1 < nb; // (no text in front of b0: the file's first record begins the job)
    if (cut_head) HIPCHK(launch_fastq_first_start(s->d_inf.as<uint8_t>(), (uint32_t)ntext, (uint32_t)(u_b0 - u_lo), 1u << 18, at_eof, d_cuts, 0, st));
    if (cut_tail) HIPCHK(launch_fastq_first_start(s->d_inf.as<uint8_t>(), (uint32_t)ntext, (uint32_t)(u_b1 - u_lo), 1u << 18, at_eof, d_cuts, 1, st));
    uint32_t* h_info = s->h_info.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(h_info, d_cuts, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_status, d_status, (size_t)nm * 4, hipMemcpyDeviceToHost, st));
    // (the first and the last byte of the text decide two small things on the host)
    HIPCHK(hipMemcpyAsync(h_info + 2, s->d_inf.as<uint8_t>(), 1, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(h_info + 2) + 1, s->d_inf.as<uint8_t>() + (ntext ? ntext - 1 : 0), 1, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s->ev, st));
    const double t_enq = ms_since(t_0);
    HIPCHK(hipEventSynchronize(s->ev));
    if (timing) fprintf(stderr, "[bgzf device] %u members, %.1f MB in, %.1f MB text: reserve %.1f ms, enqueue %.1f, wait %.1f\n", nm, cbytes / 1e6, ntext / 1e6,
                        t_reserve, t_enq - t_reserve, ms_since(t_0) - t_enq);
    for (uint32_t i = 0; i < nm; ++i) if (h_status[i] != 0) return 1;
    const uint8_t first_byte = reinterpret_cast<const uint8_t*>(h_info + 2)[0], last_byte = reinterpret_cast<const uint8_t*>(h_info + 2)[1];
    uint64_t head = cut_head ? h_info[0] : 0, tail = cut_tail ? h_info[1] : ntext;
    if (head == 0xFFFFFFFFull || tail == 0xFFFFFFFFull) return 1;
    if (head > tail) head = tail;
    if (!cut_head && tail > 0 && first_byte != '@') return 1;
    uint64_t n = tail - head;
    if (n + 1 > s->max_bytes) return 1;
    if (text_off) *text_off = u_lo + head;
    if (n == 0) return RK_OK;
    HIPCHK(hipMemcpyAsync(s->d_text.p, s->d_inf.as<uint8_t>() + head, n, hipMemcpyDeviceToDevice, st));
    if (b1 == nb && tail == ntext && last_byte != '\n') { HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + n, '\n', 1, st)); ++n; } // a last line without its newline
    HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + n, 'A', 16, st)); // the index kernels read whole 16-byte pieces
    if (!s->device_text()) HIPCHK(hipMemcpyAsync(s->h_text.p, s->d_text.p, n, hipMemcpyDeviceToHost, st)); // names, sequences and qualities for the formatters
    s->text_on_device = true;
    *nbytes = n;
    return RK_OK;
}

// The work buffers of rk_fastq_slot_load_gzip made ahead of the first call, for stretches of up to comp_bytes compressed bytes
// (rk_gzip_stretch_bytes after rk_gzip_plan).  Optional: the first call makes whatever is missing.
extern "C" int rk_fastq_slot_reserve_gzip(rk_fastq_slot* s, uint64_t comp_bytes) {
    if (!s || !s->device_text()) return fail(RK_ERR_ARG, "rk_fastq_slot_reserve_gzip: a slot created with RK_SLOT_DEVICE_TEXT is needed");
    if (!s->gzs) s->gzs = new GzScratch();
    return gzip_reserve(*s->gzs, s->c, comp_bytes, s->max_bytes);
}

// The next stretch of an ORDINARY gzip file (one deflate stream: rk_gzip_open, rk_gzip_plan) inflated on the device (rk_gunzip.hip) into
// this slot's device text: whole records, *nbytes of them (0: this stretch completed none), the first one at byte *text_off of the
// file's text.  Calls come in order, call = 0 .. rk_gzip_plan() - 1, all from slots of ONE device; the next rk_fastq_slot_submit /
// _count of this slot takes *nbytes and skips its upload.  Returns RK_OK; 1: the device route ends here -- the caller's sequential
// reader continues from *text_off (rk_reader_open_at); < 0: an error (damaged data).
extern "C" int rk_fastq_slot_load_gzip(rk_fastq_slot* s, rk_gzip* gz, int64_t call, uint64_t* nbytes, uint64_t* text_off) {
    if (!s || !gz || !nbytes || !text_off) return fail(RK_ERR_ARG, "bad arguments");
    if (!s->device_text()) return fail(RK_ERR_ARG, "rk_fastq_slot_load_gzip: a slot created with RK_SLOT_DEVICE_TEXT is needed");
    s->text_on_device = false;
    if (!s->gzs) s->gzs = new GzScratch();
    const int rc = gzip_next(gz, *s->gzs, s->c, s->st, s->ev, call, s->d_text.as<uint8_t>(), s->max_bytes, nbytes, text_off);
    if (rc != RK_OK) return rc;
    HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + *nbytes, 'A', 16, s->st)); // the index kernels read whole 16-byte pieces
    s->text_on_device = true;
    return RK_OK;
}

// The two halves of rk_fastq_slot_classify, for callers that keep two slots per thread: submit() enqueues the upload and the
// index / check / pack kernels and returns at once; finish() waits for them, launches the classification and collects the rows.
// Between the two the caller can read its next block into its other slot -- the link and the GPU work while the host reads.
static int slot_submit(rk_fastq_slot* s, uint64_t nbytes, bool for_output) {
    if (!s || nbytes > s->max_bytes) return fail(RK_ERR_ARG, "bad arguments");
    rk_ctx* c = s->c;
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    s->pending = nbytes;
    s->submitted = true;
    if (nbytes == 0) return RK_OK;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    s->src = nullptr;
    if (s->text_on_device) { s->text_on_device = false; s->next_src = nullptr; } // rk_fastq_slot_load_bgzf left this block's text in d_text
    else if (s->next_src) { // straight from the caller's (page-locked) memory: no copy into the slot's buffer
        s->src = s->next_src; s->next_src = nullptr;
        HIPCHK(hipMemcpyAsync(s->d_text.p, s->src, nbytes, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(s->d_text.as<uint8_t>() + nbytes, 'A', 16, st)); // the device reads whole 16-byte pieces
    } else {
        if (s->device_text()) return fail(RK_ERR_STATE, "a device-text slot has no host text buffer: load the block with rk_fastq_slot_load_bgzf or name it with rk_fastq_slot_set_source");
        uint8_t* text = s->h_text.as<uint8_t>();
        memset(text + nbytes, 'A', 16); // the device reads whole 16-byte pieces
        HIPCHK(hipMemcpyAsync(s->d_text.p, text, (nbytes + 15) & ~(uint64_t)15, hipMemcpyHostToDevice, st));
    }
    HIPCHK(launch_fastq_index(s->d, s->d_text.as<uint8_t>(), nbytes, st));
    // device-text slots, stream / classify: the names are packed right away (they do not depend on the rows)
    if (for_output && s->device_text() && !s->pack_filter) HIPCHK(launch_fastq_pack(s->d, s->d_text.as<uint8_t>(), true, nullptr, 0, 0, st));
    HIPCHK(hipMemcpyAsync(s->h_info.p, s->d.info, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(s->ev, st));
    return RK_OK;
}
extern "C" int rk_fastq_slot_submit(rk_fastq_slot* s, uint64_t nbytes) { return slot_submit(s, nbytes, true); }

extern "C" int rk_fastq_slot_finish(rk_fastq_slot* s, rk_fastq_result* res) {
    if (!s || !res) return fail(RK_ERR_ARG, "bad arguments");
    if (!s->submitted) return fail(RK_ERR_STATE, "rk_fastq_slot_finish without rk_fastq_slot_submit");
    s->submitted = false;
    rk_ctx* c = s->c;
    memset(res, 0, sizeof *res);
    if (s->pending == 0) return RK_OK;
    RKCHK(set_dev(c));
    hipStream_t st = s->st;
    const bool dt = s->device_text();
    s->spans_base = dt ? nullptr : (s->src ? s->src : s->h_text.as<uint8_t>());
    uint32_t* info = s->h_info.as<uint32_t>();
    HIPCHK(hipEventSynchronize(s->ev));
    if (info[0] != 0) { res->status = (int32_t)info[0]; return RK_OK; } // not strictly four lines per record: the caller's scanner takes the block
    const int64_t nrec = (int64_t)info[1];
    res->nrec = nrec;
    if (nrec == 0) return RK_OK;
    const size_t nr = (size_t)nrec;
    int32_t* out4 = static_cast<int32_t*>(s->h_out4.get(nr * 16));
    uint32_t* spans = static_cast<uint32_t*>(s->h_spans.get(nr * 20));
    RKCHK(fused_device(c, s->d.bases, s->d.out_off, nrec, s->d_out4.p, info[2], 0, nullptr, st));
    HIPCHK(hipMemcpyAsync(out4, s->d_out4.p, nr * 16, hipMemcpyDeviceToHost, st));
    uint8_t* pack = nullptr;
    if (!dt) {
        HIPCHK(hipMemcpyAsync(spans, s->d.name_off, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + nr, s->d.name_len, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + 2 * nr, s->d.seq_off, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + 3 * nr, s->d.seq_len, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + 4 * nr, s->d.qual_off, nr * 4, hipMemcpyDeviceToHost, st));
    } else if (!s->pack_filter) { // the names, packed by submit: info[3] bytes; name i = pack[pack_off[i] .. + name_len[i])
        pack = static_cast<uint8_t*>(s->h_pack.get((size_t)info[3] + 64));
        HIPCHK(hipMemcpyAsync(spans, s->d.pack_off, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + nr, s->d.name_len, nr * 4, hipMemcpyDeviceToHost, st));
        if (info[3]) HIPCHK(hipMemcpyAsync(pack, s->d.pack, info[3], hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipEventRecord(s->ev, st));
    HIPCHK(hipEventSynchronize(s->ev));
    // rows the fused kernel handed back (long reads, more windows than the sketch keeps, ...): the general path, on the packed bases where they lie
    RKCHK(reroute_flagged_device(c, s->d.bases, s->d.out_off, nrec, s->d_out4.p, out4, st));
    res->out4 = out4;
    res->name_off = spans; res->name_len = spans + nr; res->seq_off = spans + 2 * nr; res->seq_len = spans + 3 * nr; res->qual_off = spans + 4 * nr;
    if (dt && s->pack_filter) {
        // filter: the records it prints (decided on the final rows) are packed now: name, sequence, quality string back to back
        HIPCHK(launch_fastq_pack(s->d, s->d_text.as<uint8_t>(), false, s->d_out4.as<int32_t>(), s->min_matches, s->min_diff, st));
        HIPCHK(hipMemcpyAsync(info, s->d.info, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans, s->d.pack_off, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + nr, s->d.name_len, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(spans + 3 * nr, s->d.seq_len, nr * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(s->ev, st));
        HIPCHK(hipEventSynchronize(s->ev));
        if (info[0] != 0) { res->status = (int32_t)info[0]; return RK_OK; }
        pack = static_cast<uint8_t*>(s->h_pack.get((size_t)info[3] + 64));
        if (info[3]) HIPCHK(hipMemcpyAsync(pack, s->d.pack, info[3], hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(s->ev, st));
        uint32_t* so = spans + 2 * nr; uint32_t* qo = spans + 4 * nr;
        for (size_t i = 0; i < nr; ++i) { so[i] = spans[i] + spans[nr + i]; qo[i] = so[i] + spans[3 * nr + i]; } // (while the bytes travel)
        HIPCHK(hipEventSynchronize(s->ev));
    }
    if (dt) {
        memset(pack + info[3], 0, 32); // (the formatters copy names in 16-byte steps)
        s->spans_base = pack;
    }
    return RK_OK;
}

// Pass 1 of -M on a block of raw FASTQ text (rkmh.cpp:904-910): split, check and pack on the device as rk_fastq_slot_classify does,
// then count every window's hash into `counter`.  *status != 0: the block is not four lines per record and NOTHING was counted.
extern "C" int rk_fastq_slot_count(rk_fastq_slot* s, uint64_t nbytes, rk_counter* counter, int32_t* status, int64_t* nrec_out) {
    if (!s || !counter || !status) return fail(RK_ERR_ARG, "bad arguments");
    if (counter->ctx != s->c) return fail(RK_ERR_ARG, "the counter belongs to another context");
    *status = 0;
    if (nrec_out) *nrec_out = 0;
    RKCHK(slot_submit(s, nbytes, false));
    s->submitted = false;
    if (nbytes == 0) return RK_OK;
    rk_ctx* c = s->c;
    RKCHK(set_dev(c));
    uint32_t* info = s->h_info.as<uint32_t>();
    HIPCHK(hipEventSynchronize(s->ev));
    if (info[0] != 0) { *status = (int32_t)info[0]; return RK_OK; }
    const int64_t nrec = (int64_t)info[1];
    if (nrec_out) *nrec_out = nrec;
    if (nrec == 0) return RK_OK;
    if (info[2] > (uint32_t)FUSED_MAXLEN && counter->compact)
        return fail(RK_ERR_NEED_FULL, "reads longer than %d bases: a compact depth map only counts reads that fit the sketch", FUSED_MAXLEN);
    if (info[2] > (uint32_t)FUSED_MAXLEN) {
        // a read longer than the fused kernel's limit: the whole block through the tile hasher, on the packed bases where they lie
        std::vector<uint32_t> off32((size_t)nrec + 1);
        HIPCHK(hipMemcpyAsync(off32.data(), s->d.out_off, ((size_t)nrec + 1) * 4, hipMemcpyDeviceToHost, s->st));
        HIPCHK(hipEventRecord(s->ev, s->st));
        HIPCHK(hipEventSynchronize(s->ev));
        std::vector<uint64_t> lens_ps((size_t)nrec + 1, 0), starts((size_t)nrec);
        for (int64_t i = 0; i < nrec; ++i) { starts[(size_t)i] = off32[(size_t)i]; lens_ps[(size_t)i + 1] = lens_ps[(size_t)i] + (off32[(size_t)i + 1] - off32[(size_t)i]); }
        std::lock_guard<std::mutex> lock(c->general_mu);
        RKCHK(counter_settle(counter));
        GeneralCfg cfg; cfg.ks = c->ks; cfg.inc_counter = counter; cfg.abs_starts = starts.data();
        GeneralOut none;
        return general_run(c, nullptr, s->d.bases, lens_ps.data(), nrec, cfg, none);
    }
    RKCHK(fused_device(c, s->d.bases, s->d.out_off, nrec, nullptr, info[2], 1, counter, s->st));
    HIPCHK(hipEventRecord(s->ev, s->st));
    HIPCHK(hipEventSynchronize(s->ev));
    return RK_OK;
}

extern "C" int rk_fastq_slot_classify(rk_fastq_slot* s, uint64_t nbytes, rk_fastq_result* res) {
    if (!res) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(rk_fastq_slot_submit(s, nbytes));
    return rk_fastq_slot_finish(s, res);
}

// ---- reference FASTA text stripped on the device (rk_fasta.hip) -------------------------------------------------------------
struct rk_fasta_load {
    rk_ctx* c = nullptr;
    uint64_t cap = 0;
    DevBuf d_text, d_bases, d_u32, d_u64, d_rec, d_names, d_scan, d_info;
    PinBuf h_small;
    FaDev d{};
    std::vector<uint64_t> offsets, name_offsets;
    std::vector<char> names;
    int64_t nseq = 0;
    bool finished = false;
};

extern "C" void rk_fasta_load_destroy(rk_fasta_load* L) {
    if (!L) return;
    if (L->c) { hipError_t e = hipSetDevice(L->c->device); (void)e; e = hipStreamSynchronize(L->c->st); (void)e; }
    for (DevBuf* b : {&L->d_text, &L->d_bases, &L->d_u32, &L->d_u64, &L->d_rec, &L->d_names, &L->d_scan, &L->d_info}) b->release();
    L->h_small.release();
    delete L;
}

extern "C" int rk_fasta_load_create(rk_ctx* c, uint64_t text_bytes, rk_fasta_load** out) {
    if (!c || !out || text_bytes < 1 || text_bytes > ((uint64_t)1 << 37)) return fail(RK_ERR_ARG, "bad arguments (1 byte .. 128 GB of text)");
    RKCHK(set_dev(c));
    rk_fasta_load* L = new rk_fasta_load();
    L->c = c; L->cap = text_bytes;
    struct Guard { rk_fasta_load* L; ~Guard() { if (L) rk_fasta_load_destroy(L); } } guard{L};
    const uint64_t chunks = fa_chunks(text_bytes);
    RKCHK(L->d_text.reserve(chunks * 4096 + 64)); // the kernels read whole 4 KB chunks
    RKCHK(L->d_u32.reserve(2 * chunks * 4 + 64));
    RKCHK(L->d_u64.reserve(4 * (chunks + 1) * 8 + 64));
    RKCHK(L->d_info.reserve(64));
    RKCHK(L->h_small.reserve(64));
    guard.L = nullptr;
    *out = L;
    return RK_OK;
}

// the first nbytes of the slot's page-locked text buffer become text[text_offset ..); returns when the buffer may be refilled
extern "C" int rk_fasta_load_put(rk_fasta_load* L, rk_fastq_slot* via, uint64_t text_offset, uint64_t nbytes) {
    if (!L || !via || L->finished) return fail(RK_ERR_ARG, "bad arguments");
    if (via->c->device != L->c->device) return fail(RK_ERR_ARG, "the slot belongs to another device");
    if (nbytes > via->max_bytes || text_offset > L->cap || nbytes > L->cap - text_offset) return fail(RK_ERR_ARG, "block outside the text");
    if (nbytes == 0) return RK_OK;
    RKCHK(set_dev(L->c));
    HIPCHK(hipMemcpyAsync(L->d_text.as<uint8_t>() + text_offset, via->h_text.p, nbytes, hipMemcpyHostToDevice, via->st));
    HIPCHK(hipEventRecord(via->ev, via->st));
    HIPCHK(hipEventSynchronize(via->ev));
    return RK_OK;
}

// The text of an ORDINARY gzip file (rk_gzip_open: a genome as it is distributed, genome.fa.gz) inflated on the device (rk_gunzip.hip)
// straight into the load's text at text_offset; *text_bytes = its length.  Returns RK_OK; 1: the device route gave up (a second member
// behind the first, ...) -- the caller parses the references on the host; < 0: damaged data.  One call at a time per load.
extern "C" int rk_fasta_load_put_gzip(rk_fasta_load* L, rk_gzip* gz, uint64_t text_offset, uint64_t* text_bytes) {
    if (!L || !gz || !text_bytes || L->finished || text_offset > L->cap) return fail(RK_ERR_ARG, "bad arguments");
    *text_bytes = 0;
    rk_ctx* c = L->c;
    RKCHK(set_dev(c));
    // stretches that inflate to ~700 MB each (the work buffers are ~8 x a stretch's compressed bytes; they go when the file is done)
    const uint64_t per_call = std::min<uint64_t>((uint64_t)832 << 20, std::max<uint64_t>((uint64_t)8 << 20, L->cap - text_offset));
    const int64_t ncalls = rk_gzip_plan(gz, per_call);
    if (ncalls < 0) return RK_ERR_ARG;
    GzScratch S;
    hipEvent_t ev = nullptr;
    struct Guard { GzScratch& S; hipEvent_t& ev; ~Guard() { S.release(); if (ev) { hipError_t e = hipEventDestroy(ev); (void)e; } } } guard{S, ev};
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
    uint64_t done = 0;
    for (int64_t call = 0; call < ncalls; ++call) {
        uint64_t n = 0, off = 0;
        const uint64_t room = L->cap - text_offset - done;
        const int rc = gzip_next(gz, S, c, c->st, ev, call, L->d_text.as<uint8_t>() + text_offset + done, std::min<uint64_t>(room, per_call), &n, &off, true);
        if (rc != RK_OK) return rc;
        done += n;
    }
    HIPCHK(hipStreamSynchronize(c->st));
    *text_bytes = done;
    return RK_OK;
}

// The text of members [b0, b1) of a BGZF file (a bgzip'd genome) inflated on the device -- pass 1, pass 2 and the CRC-32 check of
// rk_inflate.hip, in the buffers of `via` (a slot created with RK_SLOT_DEVICE_TEXT whose max_bytes holds the members' text) -- and
// copied into the load's text at text_offset.  Returns RK_OK; 1: a member the device could not inflate or whose CRC-32 does not
// match (the caller parses the references on the host, which reports the damage).
extern "C" int rk_fasta_load_put_bgzf(rk_fasta_load* L, rk_fastq_slot* via, const rk_bgzf* z, int64_t b0, int64_t b1, uint64_t text_offset) {
    if (!L || !via || !z || L->finished || b0 < 0 || b1 <= b0 || b1 > rk_bgzf_members(z)) return fail(RK_ERR_ARG, "bad arguments");
    if (!via->device_text() || via->c->device != L->c->device) return fail(RK_ERR_ARG, "rk_fasta_load_put_bgzf: a device-text slot of the load's device is needed");
    rk_fastq_slot* s = via;
    RKCHK(set_dev(s->c));
    hipStream_t st = s->st;
    const uint32_t nm = (uint32_t)(b1 - b0);
    uint64_t f_lo = 0, f_hi = 0;
    uint32_t tot = 0, hd = 0, us = 0;
    RKCHK(rk_bgzf_member(z, b0, &f_lo, &tot, &hd, &us));
    RKCHK(rk_bgzf_member(z, b1 - 1, &f_hi, &tot, &hd, &us));
    const uint64_t cbytes = f_hi + tot - f_lo;
    const uint64_t u_lo = rk_bgzf_text_offset(z, b0), ntext = rk_bgzf_text_offset(z, b1) - u_lo;
    if (ntext > s->max_bytes || cbytes >= ((uint64_t)1 << 31) || text_offset > L->cap || ntext > L->cap - text_offset) return fail(RK_ERR_ARG, "rk_fasta_load_put_bgzf: the members' text does not fit the slot or the load");
    if (ntext == 0) return RK_OK;
    const size_t mem_bytes = (((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15) + (size_t)nm * 8 + 64;
    RKCHK(s->h_mem.reserve(mem_bytes));
    RKCHK(s->d_mem.reserve(mem_bytes));
    RKCHK(s->d_comp.reserve(cbytes + 256));
    RKCHK(s->d_inf.reserve(ntext + 64));
    InflateMember* mt = s->h_mem.as<InflateMember>();
    uint64_t scratch_dw = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        uint64_t fo = 0;
        RKCHK(rk_bgzf_member(z, b0 + i, &fo, &tot, &hd, &us));
        if (tot < hd + 8u) return fail(RK_ERR_IO, "BGZF member %lld is shorter than its header and footer", (long long)(b0 + i));
        mt[i].in_off = (uint32_t)(fo - f_lo) + hd; mt[i].in_len = tot - hd - 8;
        mt[i].out_off = (uint32_t)(rk_bgzf_text_offset(z, b0 + i) - u_lo); mt[i].out_len = us;
        mt[i].match_off = (uint32_t)scratch_dw; mt[i].pad = 0;
        scratch_dw += inflate_scratch_dwords(us);
    }
    if (scratch_dw >= ((uint64_t)1 << 32)) return 1;
    RKCHK(s->d_match.reserve(scratch_dw * 4 + 64));
    HIPCHK(hipMemcpyAsync(s->d_comp.p, rk_bgzf_image(z) + f_lo, cbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(s->d_comp.as<uint8_t>() + cbytes, 0, 160, st));
    const size_t cpad = ((cbytes + 15) & ~(size_t)15) + 64;
    HIPCHK(hipMemcpyAsync(s->d_mem.p, mt, (size_t)nm * sizeof(InflateMember), hipMemcpyHostToDevice, st));
    const size_t status_at = ((size_t)nm * sizeof(InflateMember) + 15) & ~(size_t)15;
    uint32_t* d_status = reinterpret_cast<uint32_t*>(s->d_mem.as<uint8_t>() + status_at);
    uint32_t* h_status = reinterpret_cast<uint32_t*>(s->h_mem.as<uint8_t>() + status_at);
    HIPCHK(launch_inflate_members(s->d_comp.as<uint8_t>(), (uint32_t)cpad, s->d_mem.as<InflateMember>(), nm, s->d_inf.as<uint8_t>(), s->d_match.as<uint32_t>(), d_status, st));
    HIPCHK(hipMemcpyAsync(h_status, d_status, (size_t)nm * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(L->d_text.as<uint8_t>() + text_offset, s->d_inf.p, ntext, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipEventRecord(s->ev, st));
    HIPCHK(hipEventSynchronize(s->ev));
    for (uint32_t i = 0; i < nm; ++i) if (h_status[i] != 0) return 1;
    return RK_OK;
}

// one '\n' into the load's text (the separator behind a file whose text was written by rk_fasta_load_put_gzip)
extern "C" int rk_fasta_load_put_newline(rk_fasta_load* L, uint64_t text_offset) {
    if (!L || L->finished || text_offset >= L->cap) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(L->c));
    HIPCHK(hipMemsetAsync(L->d_text.as<uint8_t>() + text_offset, '\n', 1, L->c->st));
    HIPCHK(hipStreamSynchronize(L->c->st));
    return RK_OK;
}

extern "C" int rk_fasta_load_finish(rk_fasta_load* L, uint64_t total_bytes, rk_fasta_index* out) {
    if (!L || !out || total_bytes < 1 || total_bytes > L->cap || L->finished) return fail(RK_ERR_ARG, "bad arguments");
    memset(out, 0, sizeof *out);
    rk_ctx* c = L->c;
    RKCHK(set_dev(c));
    hipStream_t st = c->st;
    const uint64_t chunks = fa_chunks(total_bytes);
    FaDev& d = L->d;
    d.chunk_map = L->d_u32.as<uint32_t>(); d.chunk_pre = d.chunk_map + chunks;
    d.chunk_kept = L->d_u64.as<uint64_t>(); d.chunk_hdrs = d.chunk_kept + (chunks + 1);
    d.kept_base = d.chunk_hdrs + (chunks + 1); d.hdr_base = d.kept_base + (chunks + 1);
    d.info = L->d_info.as<uint32_t>();
    RKCHK(L->d_scan.reserve(fa_scan_temp_bytes(chunks + 1)));
    d.scan_tmp = L->d_scan.p; d.scan_tmp_bytes = L->d_scan.cap;
    const uint8_t* raw = L->d_text.as<uint8_t>();
    HIPCHK(launch_fasta_count(d, raw, total_bytes, st));
    uint64_t* hs = L->h_small.as<uint64_t>(); // [0] bases, [1] records, [2] status word, [3] name bytes, [4] offset of the first record
    HIPCHK(hipMemcpyAsync(hs, d.kept_base + chunks, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 1, d.hdr_base + chunks, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 2, d.info, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const uint64_t total = hs[0], nrec = hs[1];
    uint32_t status = (uint32_t)hs[2];
    if (nrec == 0) status |= FA_BAD_EMPTY;
    if (nrec >= 0x7ffffff0ull) return fail(RK_ERR_LIMIT, "more than 2^31 reference sequences");
    if (status) { out->status = (int32_t)status; return RK_OK; }
    RKCHK(L->d_bases.reserve(total + 64));
    RKCHK(L->d_rec.reserve((4 * (nrec + 1)) * 8 + 64));
    RKCHK(L->d_scan.reserve(fa_scan_temp_bytes(nrec + 1)));
    d.scan_tmp = L->d_scan.p; d.scan_tmp_bytes = L->d_scan.cap;
    d.bases = L->d_bases.as<uint8_t>();
    d.hdr_pos = L->d_rec.as<uint64_t>(); d.rec_off = d.hdr_pos + (nrec + 1);
    d.name_len1 = d.rec_off + (nrec + 1); d.name_off = d.name_len1 + (nrec + 1);
    HIPCHK(launch_fasta_compact(d, raw, total_bytes, nrec, st));
    HIPCHK(hipMemcpyAsync(hs + 2, d.info, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 3, d.name_off + nrec, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hs + 4, d.rec_off, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    status = (uint32_t)hs[2];
    if (hs[4] != 0) status |= FA_BAD_LEAD; // bases before the first header line
    if (status) { out->status = (int32_t)status; return RK_OK; }
    const uint64_t name_bytes = hs[3];
    RKCHK(L->d_names.reserve(name_bytes + 64));
    d.names = L->d_names.as<uint8_t>();
    HIPCHK(launch_fasta_names(d, raw, nrec, st));
    L->offsets.assign((size_t)nrec + 1, 0);
    L->name_offsets.assign((size_t)nrec + 1, 0);
    L->names.assign((size_t)name_bytes + 1, 0);
    HIPCHK(hipMemcpyAsync(L->offsets.data(), d.rec_off, nrec * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(L->name_offsets.data(), d.name_off, (nrec + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(L->names.data(), d.names, name_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    L->offsets[(size_t)nrec] = total;
    L->nseq = (int64_t)nrec;
    L->finished = true;
    // the text has done its work; the packed bases stay for rk_set_references_fasta
    L->d_text.release();
    out->nseq = L->nseq;
    out->offsets = L->offsets.data();
    out->names = L->names.data();
    out->name_offsets = L->name_offsets.data();
    return RK_OK;
}

// the packed bases (offsets[nseq] bytes, as the text spells them: not upper-cased) for callers that also want them on the host
extern "C" int rk_fasta_load_get_bases(rk_fasta_load* L, uint8_t* dst) {
    if (!L || !dst || !L->finished) return fail(RK_ERR_ARG, "rk_fasta_load_get_bases needs a finished, regular rk_fasta_load");
    RKCHK(set_dev(L->c));
    const uint64_t total = L->offsets.back();
    if (total) HIPCHK(hipMemcpyAsync(dst, L->d_bases.p, total, hipMemcpyDeviceToHost, L->c->st));
    HIPCHK(hipStreamSynchronize(L->c->st));
    return RK_OK;
}

extern "C" int rk_set_references_fasta(rk_ctx* c, rk_fasta_load* L, const int* ks, int nks, int S, int max_samples, uint64_t counter_slots) {
    if (!c || !L || !L->finished) return fail(RK_ERR_ARG, "rk_set_references_fasta needs a finished, regular rk_fasta_load");
    if (L->c != c) return fail(RK_ERR_ARG, "the text was loaded through another context");
    if (L->nseq > 0x7fffffffll) return fail(RK_ERR_LIMIT, "too many reference sequences");
    return set_references_impl(c, nullptr, L->d_bases.as<uint8_t>(), L->offsets.data(), (int)L->nseq, ks, nks, S, max_samples, counter_slots);
}
