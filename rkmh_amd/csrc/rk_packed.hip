// rk_packed.hip -- packed reads on the device (include/rkmh_amd.h, "PACKED READS"): 2 bits per base cross the link, the ASCII bases the
// kernels read are rebuilt in HBM -- 16 bases per thread, one dword in, one 16-byte store out, the few bytes that are not ACGT put back
// from the exception list -- and handed to the same entry points the ASCII path uses (rk_classify_batch_device_all /
// rk_count_batch_device): the rows cannot differ.  The reference's -F/--pre-reads is parsed and unused (/root/reference/src/rkmh.cpp:659-664).
#include "rk_api_internal.hpp"

namespace rk {
namespace {

// out[16 t .. 16 t + 16) = the letters of dword t of the 2-bit stream (A 0, C 1, T 2, G 3)
__global__ __launch_bounds__(256) void k_unpack_bases(const uint32_t* __restrict__ bases2, uint64_t ndw, uint4* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ndw) return;
    const uint32_t w = bases2[t];
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t b = (w >> (8 * q)) & 0xFFu; // four bases
        // code -> letter: a byte permute over the constant "ACTG" (selector byte j = code j)
        const uint32_t sel = (b & 3u) | (((b >> 2) & 3u) << 8) | (((b >> 4) & 3u) << 16) | (((b >> 6) & 3u) << 24);
        o[q] = __builtin_amdgcn_perm(0u, 0x47544341u, sel);
    }
    out[t] = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void k_unpack_exceptions(const uint2* __restrict__ exc, uint32_t nexc, uint64_t nbases, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nexc) return;
    const uint2 e = exc[i];
    if ((uint64_t)e.x < nbases) out[e.x] = (uint8_t)e.y;
}

} // namespace
} // namespace rk

// the batch's ASCII bases in the caller's scratch buffer
static int unpack_to_ascii(const void* d_bases2, uint64_t nbases, const void* d_exc, uint32_t nexc, void* d_ascii, hipStream_t st) {
    if (((uintptr_t)d_bases2 & 3) != 0 || ((uintptr_t)d_ascii & 15) != 0) return fail(RK_ERR_ARG, "d_bases2 must be 4-byte, d_ascii 16-byte aligned");
    if (nexc && !d_exc) return fail(RK_ERR_ARG, "exceptions announced but not given");
    const uint64_t ndw = (nbases + 15) / 16;
    if (ndw) hipLaunchKernelGGL(rk::k_unpack_bases, dim3((unsigned)((ndw + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_bases2, ndw, (uint4*)d_ascii);
    if (nexc) hipLaunchKernelGGL(rk::k_unpack_exceptions, dim3((nexc + 255) / 256), dim3(256), 0, st, (const uint2*)d_exc, nexc, nbases, (uint8_t*)d_ascii);
    HIPCHK(hipGetLastError());
    return RK_OK;
}

extern "C" int rk_classify_batch_device_packed(rk_ctx* c, const void* d_bases2, const void* d_offs, int64_t nreads, uint64_t nbases, const void* d_exc,
                                               uint32_t nexc, void* d_ascii, void* d_out4, uint32_t max_read_len, void* hip_stream) {
    if (!c || nreads < 0 || (nreads > 0 && (!d_bases2 || !d_offs || !d_out4 || !d_ascii))) return fail(RK_ERR_ARG, "bad arguments");
    if (!c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    RKCHK(set_dev(c));
    if (nreads == 0) return RK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    RKCHK(unpack_to_ascii(d_bases2, nbases, d_exc, nexc, d_ascii, st));
    return rk_classify_batch_device_all(c, d_ascii, d_offs, nreads, d_out4, max_read_len, hip_stream);
}

extern "C" int rk_count_batch_device_packed(rk_ctx* c, const void* d_bases2, const void* d_offs, int64_t nreads, uint64_t nbases, const void* d_exc,
                                            uint32_t nexc, void* d_ascii, rk_counter* counter, uint32_t max_read_len, void* hip_stream) {
    if (!c || !counter || nreads < 0 || (nreads > 0 && (!d_bases2 || !d_offs || !d_ascii))) return fail(RK_ERR_ARG, "bad arguments");
    if (max_read_len == 0) return fail(RK_ERR_ARG, "rk_count_batch_device_packed: max_read_len is needed (the packed file's directory has it)");
    if (counter->ctx != c) return fail(RK_ERR_ARG, "the counter belongs to another context");
    if (c->ks.n == 0) return fail(RK_ERR_STATE, "k-mer sizes unknown: call rk_set_references first");
    RKCHK(set_dev(c));
    if (nreads == 0) return RK_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    RKCHK(unpack_to_ascii(d_bases2, nbases, d_exc, nexc, d_ascii, st));
    if (max_read_len > (uint32_t)FUSED_MAXLEN && counter->compact)
        return fail(RK_ERR_NEED_FULL, "reads longer than %d bases: a compact depth map only counts reads that fit the sketch", FUSED_MAXLEN);
    if (max_read_len > (uint32_t)FUSED_MAXLEN) { // the tile hasher, on the expanded bases where they lie (as rk_fastq_slot_count does)
        std::vector<uint32_t> off32((size_t)nreads + 1);
        HIPCHK(hipMemcpyAsync(off32.data(), d_offs, ((size_t)nreads + 1) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<uint64_t> lens_ps((size_t)nreads + 1, 0), starts((size_t)nreads);
        for (int64_t i = 0; i < nreads; ++i) { starts[(size_t)i] = off32[(size_t)i]; lens_ps[(size_t)i + 1] = lens_ps[(size_t)i] + (off32[(size_t)i + 1] - off32[(size_t)i]); }
        std::lock_guard<std::mutex> lock(c->general_mu);
        RKCHK(counter_settle(counter));
        GeneralCfg cfg; cfg.ks = c->ks; cfg.inc_counter = counter; cfg.abs_starts = starts.data();
        GeneralOut none;
        return general_run(c, nullptr, (const uint8_t*)d_ascii, lens_ps.data(), nreads, cfg, none);
    }
    return fused_device(c, d_ascii, d_offs, nreads, nullptr, max_read_len, 1, counter, st, nbases);
}

// ------------------------------------------------------------------------------------------------
// One block of a packed file in flight (own stream, device arrays, page-locked rows): the unit `stream|filter -F` works with.
struct rk_packed_slot {
    rk_ctx* c = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr; // (blocking-sync: the waiting threads sleep)
    uint64_t max_reads = 0, max_bases = 0;
    DevBuf d_offs, d_bases2, d_exc, d_ascii, d_out4;
    PinBuf h_out4;
    std::vector<uint32_t> name_len, seq_len;
};
extern "C" void rk_packed_slot_destroy(rk_packed_slot* s) {
    if (!s) return;
    if (s->c) { hipError_t e = hipSetDevice(s->c->device); (void)e; }
    if (s->st) { hipError_t e = hipStreamSynchronize(s->st); (void)e; e = hipStreamDestroy(s->st); (void)e; }
    if (s->ev) { hipError_t e = hipEventDestroy(s->ev); (void)e; }
    for (DevBuf* b : {&s->d_offs, &s->d_bases2, &s->d_exc, &s->d_ascii, &s->d_out4}) b->release();
    s->h_out4.release();
    delete s;
}
extern "C" int rk_packed_slot_create(rk_ctx* c, uint64_t max_reads, uint64_t max_bases, rk_packed_slot** out) {
    if (!c || !out || max_reads < 1 || max_reads > 0x7ffffff0ull || max_bases > ((uint64_t)1 << 32) - 64) return fail(RK_ERR_ARG, "bad arguments");
    RKCHK(set_dev(c));
    rk_packed_slot* s = new rk_packed_slot();
    s->c = c; s->max_reads = max_reads; s->max_bases = max_bases;
    struct Guard { rk_packed_slot* s; ~Guard() { if (s) rk_packed_slot_destroy(s); } } guard{s};
    HIPCHK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->ev, hipEventBlockingSync | hipEventDisableTiming));
    RKCHK(s->d_offs.reserve((max_reads + 1) * 4 + 64));
    RKCHK(s->d_bases2.reserve(max_bases / 4 + 64));
    RKCHK(s->d_ascii.reserve(max_bases + 128));
    RKCHK(s->d_out4.reserve(max_reads * 16));
    RKCHK(s->h_out4.reserve(max_reads * 16));
    guard.s = nullptr;
    *out = s;
    return RK_OK;
}
static int packed_upload(rk_packed_slot* s, const rk_packed_block* b, const uint8_t* file) {
    if (!s || !b || !file) return fail(RK_ERR_ARG, "bad arguments");
    if (b->nrec > s->max_reads || b->nbases > s->max_bases) return fail(RK_ERR_LIMIT, "the block (%u reads, %llu bases) is larger than the slot", b->nrec, (unsigned long long)b->nbases);
    RKCHK(set_dev(s->c));
    if (b->nexc) RKCHK(s->d_exc.reserve((size_t)b->nexc * 8));
    HIPCHK(hipMemcpyAsync(s->d_offs.p, file + b->offsets_off, ((size_t)b->nrec + 1) * 4, hipMemcpyHostToDevice, s->st));
    HIPCHK(hipMemcpyAsync(s->d_bases2.p, file + b->bases_off, (size_t)((b->nbases + 3) / 4), hipMemcpyHostToDevice, s->st));
    if (b->nexc) HIPCHK(hipMemcpyAsync(s->d_exc.p, file + b->exc_off, (size_t)b->nexc * 8, hipMemcpyHostToDevice, s->st));
    return RK_OK;
}
// The block's sections are read from `file` (the mapped packed file; page-locked with rk_host_register_readonly the DMA engine reads
// the page cache itself); res->out4 = the rows, name_off / name_len index file + block->names_off, seq_off (= qual_off) / seq_len are in
// BASES from the block's first base.  The arrays live in the slot (and in the mapping) until its next call.
extern "C" int rk_packed_slot_classify(rk_packed_slot* s, const rk_packed_block* b, const uint8_t* file, rk_fastq_result* res) {
    if (!res) return fail(RK_ERR_ARG, "bad arguments");
    memset(res, 0, sizeof *res);
    RKCHK(packed_upload(s, b, file));
    res->nrec = b->nrec;
    if (b->nrec == 0) return RK_OK;
    if (!s->c->have_refs) return fail(RK_ERR_STATE, "classify before rk_set_references");
    RKCHK(unpack_to_ascii(s->d_bases2.p, b->nbases, b->nexc ? s->d_exc.p : nullptr, b->nexc, s->d_ascii.p, s->st));
    RKCHK(rk_classify_batch_device(s->c, s->d_ascii.p, s->d_offs.p, b->nrec, s->d_out4.p, b->max_len ? b->max_len : 1u, s->st));
    HIPCHK(hipMemcpyAsync(s->h_out4.p, s->d_out4.p, (size_t)b->nrec * 16, hipMemcpyDeviceToHost, s->st));
    HIPCHK(hipEventRecord(s->ev, s->st));
    const uint32_t* no = reinterpret_cast<const uint32_t*>(file + b->name_offsets_off);
    const uint32_t* so = reinterpret_cast<const uint32_t*>(file + b->offsets_off);
    s->name_len.resize(b->nrec); s->seq_len.resize(b->nrec);
    for (uint32_t i = 0; i < b->nrec; ++i) { s->name_len[i] = no[i + 1] - no[i]; s->seq_len[i] = so[i + 1] - so[i]; } // (while the rows travel)
    HIPCHK(hipEventSynchronize(s->ev));
    // rows the fused kernel handed back (long reads, more windows than the sketch keeps): the general kernels, on the expanded bases
    RKCHK(reroute_flagged_device(s->c, s->d_ascii.p, s->d_offs.p, b->nrec, s->d_out4.p, s->h_out4.as<int32_t>(), s->st));
    res->out4 = s->h_out4.as<int32_t>();
    res->name_off = no; res->name_len = s->name_len.data();
    res->seq_off = so; res->seq_len = s->seq_len.data(); res->qual_off = so;
    return RK_OK;
}
extern "C" int rk_packed_slot_count(rk_packed_slot* s, const rk_packed_block* b, const uint8_t* file, rk_counter* counter) {
    RKCHK(packed_upload(s, b, file));
    if (b->nrec == 0) return RK_OK;
    RKCHK(rk_count_batch_device_packed(s->c, s->d_bases2.p, s->d_offs.p, b->nrec, b->nbases, b->nexc ? s->d_exc.p : nullptr, b->nexc, s->d_ascii.p, counter,
                                       b->max_len ? b->max_len : 1u, s->st));
    HIPCHK(hipEventRecord(s->ev, s->st));
    HIPCHK(hipEventSynchronize(s->ev));
    return RK_OK;
}
