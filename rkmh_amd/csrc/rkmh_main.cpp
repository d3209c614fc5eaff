// rkmh_main.cpp -- the `rkmh` command line on top of librkmh_amd.so (C ABI in include/rkmh_amd.h).
//
// Drop-in for the sub-commands of /root/reference/src/rkmh.cpp that sit on the classify/stream hot path:
//   stream / classify   main_stream   (src/rkmh.cpp:584-989; classify forwards to it, :2744-2747)
//   filter              main_filter   (src/rkmh.cpp:996-1424)
//   call                main_call     (src/rkmh.cpp:1455-1904)
//   hash                main_hash     (src/rkmh.cpp:1931-2116)
// Same flags (option tables src/rkmh.cpp:626-650 and :1963-1983), same stdout line formats
// (src/rkmh.cpp:892), but the per-read OpenMP loop is replaced by batches handed to the GPU while a
// second host thread parses the next batch.  Output order = input order (the reference's order is
// nondeterministic under -t > 1, src/rkmh.cpp:893).
#include <getopt.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rkmh_amd.h"

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <chrono>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_timing = getenv("RKMH_TIMING") != nullptr; // stage timings on stderr
static void tick(const char* what, double& t0) {
    if (!g_timing) return;
    const double t = now_s();
    fprintf(stderr, "[rkmh timing] %-28s %.3f s\n", what, t - t0);
    t0 = t;
}

// The command returns when its OUTPUT is complete, not when the kernel has finished taking the process apart.  After _exit the
// driver still unpins every page-locked buffer, unmaps the queues and frees the device memory of the process -- 0.2 to 0.7 s that grew
// with the run (profiles/r05_c3_e2e.txt: "the process leaving"), as much as the main loop of a 64 M-read run.  So main() forks before
// anything touches the GPU: the child does the work, and once every byte is written it closes its output descriptors, tells the parent
// its exit status through a pipe and leaves; the parent -- which holds no GPU state at all -- exits with that status at once, and the
// child's teardown runs on behind it.  A child that dies any other way is waited for and its status passed on.  Not under a
// profiler or RKMH_SLOW_EXIT=1 (the orderly way out), nor with RKMH_FORK=0.
static int g_done_fd = -1; // (child) write end of the status pipe
static void tell_parent(int status) {
    if (g_done_fd < 0) return;
    fflush(stdout); fflush(stderr);
    // Standard output a regular file: every byte is in the page cache, the parent may go -- and only then is the file closed: this is
    // the last descriptor of it (the parent closed its copy after the fork), and ext4 starts allocating and writing back a file that
    // was opened with O_TRUNC ("> out.tsv") at its last close (auto_da_alloc): ~0.1 s per GB, 0.45 s of a 100 M-read run's wall
    // clock (profiles/r06_c3_e2e.txt), which no reader of the file waits for.  A pipe or a terminal: closed first, so that a reader
    // sees the end of the stream no later than the command's return.
    struct stat st;
    const bool regular = fstat(1, &st) == 0 && S_ISREG(st.st_mode);
    if (!regular) { close(1); close(2); }
    const unsigned char b = (unsigned char)status;
    if (write(g_done_fd, &b, 1) != 1) {}
    close(g_done_fd);
    g_done_fd = -1;
    if (regular) { close(1); close(2); }
}
// Leaving after an error: flush what there is and go, WITHOUT running static destructors -- a parser or worker thread may still be
// running, and the HIP runtime's exit handlers are not something to run under it.
[[noreturn]] static void fail_exit() {
    fflush(stdout); fflush(stderr);
    tell_parent(1);
    _exit(1);
}
static void die(const char* what) {
    fprintf(stderr, "rkmh: %s: %s\n", what, rk_last_error());
    fail_exit();
}
// A profiler's tool library has initialised the GPU runtime before main() (a forked child could not use it) and writes its tables
// from an exit handler (so the process must leave through exit()): rocprofv3 / rocprof / roctracer announce themselves through
// ROCP* / HSA_TOOLS_LIB variables or a preloaded library of theirs.  (Any OTHER preloaded library -- a sanitizer, an exec guard --
// is no reason to give up the fast exit: a first form tested LD_PRELOAD alone, and on a machine that preloads a guard library into
// every process the fork never happened.)
extern char** environ;
static bool under_profiler() {
    static const bool yes = [] {
        for (char** e = environ; e && *e; ++e)
            if (strncmp(*e, "ROCP", 4) == 0 || strncmp(*e, "HSA_TOOLS_LIB=", 14) == 0) return true;
        const char* pre = getenv("LD_PRELOAD");
        return pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "rocsys") || strstr(pre, "omnitrace") || strstr(pre, "omniperf"));
    }();
    return yes;
}
// Leaving after success: only if every byte really reached standard output (a full disk or a closed pipe must not exit 0)
static const double g_loaded_s = now_s(); // (static initialisation: the program and its libraries are loaded)
[[noreturn]] static void done_exit() {
    if (g_timing) {
        fprintf(stderr, "[rkmh timing] %-28s %.3f s\n", "since the program was loaded", now_s() - g_loaded_s);
        // (for scripts that bracket the command with `date +%s.%N`: where the wall clock outside the program goes -- before it was loaded or after its last line)
        const double epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[rkmh timing] loaded at epoch %.3f, leaving at epoch %.3f\n", epoch - (now_s() - g_loaded_s), epoch);
    }
    const bool bad = fflush(stdout) != 0 || ferror(stdout);
    fflush(stderr);
    if (bad) fprintf(stderr, "rkmh: write error on standard output\n");
    if (getenv("RKMH_SLOW_EXIT") || under_profiler()) exit(bad ? 1 : 0); // profilers (rocprofv3) write their tables from an exit handler
    tell_parent(bad ? 1 : 0);
    _exit(bad ? 1 : 0); // skips the HIP runtime's and the loader's exit handlers (~0.1-0.2 s of a 1 s run)
}
#define CK(call) do { if ((call) != RK_OK) die(#call); } while (0)

// The hashing policy of this run: the build's defaults, then RKMH_POLICY, then --hash-policy (rk_policy_parse: presets `default`
// and `mash`, or fold= / windows= / zero= / mask= / freqmax= / seed=).  The arithmetic behind these switches is mkmh's, which the
// reference's tree does not hold (src/rkmh.cpp:17); every context of the process is created with g_policy.
static rk_policy g_policy;
static void policy_apply(const char* spec, const char* from) {
    if (rk_policy_parse(spec, &g_policy) != RK_OK) { fprintf(stderr, "rkmh: %s: %s\n", from, rk_last_error()); exit(1); }
}
static std::string policy_text(const rk_policy& p) {
    char b[160];
    if (rk_policy_describe(&p, b, sizeof b) < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); exit(1); }
    return b;
}
#define HASH_POLICY_OPTION {"hash-policy", required_argument, 0, 1004}
#define HASH_POLICY_HELP \
    "  --hash-policy <spec>    the mkmh choices the reference's tree does not fix, as presets (default, mash) and/or key=value:\n" \
    "                          fold=swap32|h1|w2w1, windows=len-k|len-k+1, zero=count|skip, mask=lt|le, freqmax=incl|excl, seed=<n>;\n" \
    "                          `mash` = fold=h1,windows=len-k+1 (the sketches Mash / sourmash compute).  RKMH_POLICY: the same, read first\n"

static void print_help() {
    fprintf(stderr,
            "rkmh (MI355X build): MinHash read classification on AMD Instinct GPUs\n"
            "Usage: rkmh <command> [options]\n"
            "  classify / stream   classify reads against a set of references\n"
            "  filter              print the reads that match a reference (or classify reads arriving on STDIN)\n"
            "  call                call SNPs / 1-bp deletions from k-mer depth along a reference\n"
            "  hash                print the k-mer hashes of every sequence\n"
            "  hpv16               HPV type and HPV16 lineage / sublineage k-mer matches of every read\n"
            "  sketch              write MinHash sketches as JSON (load them with stream -R)\n"
            "  pack                write reads as a packed file (2 bits per base + names): stream|filter -F <file> classifies it without parsing\n"
            "Run a command without options for its help text.\n");
}
static void help_stream() {
    fprintf(stderr,
            "rkmh stream|classify -r <refs.fa> -f <reads.fq> [-k <k>]... [-s <sketch>] [options]\n"
            "  -r/--reference <file>   reference FASTA/FASTQ(.gz); repeatable\n"
            "  -f/--fasta <file>       read FASTA/FASTQ(.gz); repeatable\n"
            "  -k/--kmer <k>           k-mer size; repeatable (default 16)\n"
            "  -s/--sketch-size <s>    sketch size (default 1000; at most 16384 in this build)\n"
            "  -t/--threads <n>        accepted for compatibility (the per-read loop runs on the GPU)\n"
            "  -M/--min-kmer-occurence <n>  drop read k-mers seen fewer than n times across all reads\n"
            "  -I/--max-samples <n>    drop reference k-mers counted more than n times across references\n"
            "  -N/--min-matches <n>    flag FAIL:DEPTH / FAIL:MATCHES\n"
            "  -D/--min-diff <n>       flag FAIL:DIFF\n"
            "  -p/-q <file>, -S <n>, -i, -z, -m   parsed and ignored, as in the reference\n"
            "  -F/--pre-reads <file.rkp>  reads packed by `rkmh pack` (2 bits per base + names) instead of -f text; repeatable\n"
            "  -R <sketches.json>      reference sketches written by `rkmh sketch` instead of -r\n"
            "  --depth-map-cache <file>  (with -M) save the read-depth map of this run, or reuse the file if it was saved\n"
            "                          from the same reads, k-mer sizes and hashing policy (anything else is refused)\n"
            "  --kmer-cache <file>       keep the k-mer enumeration of these references (k 8 .. 18; up to 20 once the file exists) in\n"
            "                          <file>; reused while references, k and hashing policy match.  Without it, ONE k of 17 .. 20 keeps\n"
            "                          its enumeration in <first -r file>.k<k>.s<s>.rkkc (--no-kmer-cache: not; k 19 / 20 then hash every window)\n"
            HASH_POLICY_HELP
            "  --device <id>           GPU to use (default 0)\n"
            "  --devices <a,b,..|all>  spread the reads over several GPUs of this node (stream, filter): one host thread and one\n"
            "                          context per device, reference sketches built on the first and imported by the others, -M depth\n"
            "                          tables summed after pass 1; output order and content are those of a single-device run\n");
}
static void help_hash() {
    fprintf(stderr,
            "rkmh hash -f <seqs.fa|fq> [-k <k>]... [--hash-policy <spec>]\n"
            "  prints one line per sequence: name, then every k-mer hash, tab separated\n" HASH_POLICY_HELP);
}

struct Opts {
    std::vector<const char*> refs, reads;
    std::vector<const char*> packed; // -F <file>: reads written by `rkmh pack`
    std::vector<int> ks;
    int sketch = 1000, threads = 1, min_occ = -1, min_matches = -1, min_diff = 0, max_samples = 100000;
    const char* kmer_cache = getenv("RKMH_KMER_CACHE"); // --kmer-cache FILE: the k-mer enumeration of these references, kept between runs (rk_set_kmer_cache)
    bool read_depth = false, ref_depth = false;
    int device = 0;
    std::vector<int> devices; // --devices a,b,...: reads are spread over these GPUs (one host thread + rk_ctx each); empty = --device
};
static std::vector<int> parse_devices(const char* arg) {
    std::vector<int> d;
    if (!strcmp(arg, "all")) { const int n = rk_device_count(); for (int i = 0; i < n; ++i) d.push_back(i); return d; }
    for (const char* p = arg; *p;) {
        char* e = nullptr;
        const long v = strtol(p, &e, 10);
        if (e == p || v < 0) { fprintf(stderr, "rkmh: bad --devices list '%s'\n", arg); exit(1); }
        d.push_back((int)v);
        p = *e == ',' ? e + 1 : e;
        if (*e && *e != ',') { fprintf(stderr, "rkmh: bad --devices list '%s'\n", arg); exit(1); }
    }
    return d;
}

// bounded queue between pipeline stages (parser -> classify -> format/write)
template <typename V> struct QueueT {
    std::mutex m;
    std::condition_variable cv;
    std::deque<V> q;
    bool done = false;
    size_t cap = 2;
    std::string err;
    void push(V s) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.size() < cap; });
        q.push_back(std::move(s));
        cv.notify_all();
    }
    bool pop(V* s) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || done; });
        if (q.empty()) return false;
        *s = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    // the next element and, in the same step, up to maxn - 1 more that are queued right now: a run no other consumer can cut into
    // (maxn_of: how many at most, given the first one)
    template <typename F> bool pop_run(std::vector<V>* run, F maxn_of) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || done; });
        run->clear();
        const size_t maxn = q.empty() ? 0 : (size_t)maxn_of(q.front());
        while (!q.empty() && run->size() < maxn) { run->push_back(std::move(q.front())); q.pop_front(); }
        cv.notify_all();
        return !run->empty();
    }
    void finish() { std::lock_guard<std::mutex> l(m); done = true; cv.notify_all(); }
};
typedef QueueT<rk_seqset> Queue;
// Result rows of the streaming path live in a few recycled page-locked buffers (rk_host_alloc): a fresh 16 MB vector per batch is
// zero-filled and page-faulted by the host each time, and the library would have to page-lock or stage it (8 of the 14 ms a
// 1 M-read batch spent in its classify stage)
struct OutPool {
    std::mutex m;
    std::vector<std::pair<int32_t*, size_t>> free_; // (buffer, rows it holds)
    int32_t* get(size_t rows, size_t* cap) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].second >= rows) { int32_t* p = free_[i].first; *cap = free_[i].second; free_.erase(free_.begin() + (long)i); return p; }
        }
        size_t want = rows + rows / 4 + 4096;
        void* p = nullptr;
        if (rk_host_alloc(want * 16, &p) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); exit(1); }
        *cap = want;
        return (int32_t*)p;
    }
    void put(int32_t* p, size_t cap) { if (p) { std::lock_guard<std::mutex> l(m); free_.emplace_back(p, cap); } }
    ~OutPool() { for (auto& f : free_) rk_host_free(f.first); }
};
struct Classified { rk_seqset reads; int32_t* out4 = nullptr; size_t out_cap = 0; int64_t seq = 0; };
struct Numbered { rk_seqset reads; int64_t seq = 0; };

static inline char* put_int(char* w, int v) {
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *w++ = '-';
    while (n) *w++ = tmp[--n];
    return w;
}

// Lines of reads [lo, hi) in the format of rk_format_stream_line (rkmh.cpp:887-892), written straight into one buffer:
// name lengths come from the offset arrays, no per-line strlen / temporary / append.
static void format_range(const rk_seqset& refs, const rk_seqset& reads, const int32_t* out4, const Opts& o,
                         int64_t lo, int64_t hi, std::string& buf) {
    size_t maxref = 0;
    for (int64_t r = 0; r < refs.nseq; ++r) maxref = std::max<size_t>(maxref, (size_t)(refs.name_offsets[r + 1] - refs.name_offsets[r]));
    const size_t need = (size_t)(reads.name_offsets[hi] - reads.name_offsets[lo]) + (size_t)(hi - lo) * (maxref + 64);
    if (buf.size() < need) buf.resize(need);
    char* const w0 = &buf[0];
    char* w = w0;
    for (int64_t i = lo; i < hi; ++i) {
        const int32_t* r = out4 + i * 4;
        const size_t ln = (size_t)(refs.name_offsets[r[0] + 1] - refs.name_offsets[r[0]]) - 1; // offsets include the NUL
        const size_t lq = (size_t)(reads.name_offsets[i + 1] - reads.name_offsets[i]) - 1;
        memcpy(w, refs.names + refs.name_offsets[r[0]], ln); w += ln; *w++ = '\t';
        memcpy(w, reads.names + reads.name_offsets[i], lq); w += lq; *w++ = '\t';
        w = put_int(w, r[1]); *w++ = '\t';
        w = put_int(w, o.sketch);
        if (r[3] <= o.min_matches) { memcpy(w, "FAIL:DEPTH", 10); w += 10; }
        *w++ = '\t';
        if (r[1] < o.min_matches) { memcpy(w, "FAIL:MATCHES", 12); w += 12; }
        *w++ = '\t';
        if (!(r[2] > o.min_diff)) { memcpy(w, "FAIL:DIFF", 9); w += 9; }
        *w++ = '\n';
    }
    buf.resize((size_t)(w - w0));
}

// TSV lines in read order (rkmh.cpp:889-897); big batches are formatted by a few threads, written in order
static void emit_lines(const rk_seqset& refs, const rk_seqset& reads, const int32_t* out4, const Opts& o, std::string& buf) {
    const int nt = reads.nseq >= 65536 ? 6 : 1;
    if (nt == 1) {
        format_range(refs, reads, out4, o, 0, reads.nseq, buf);
        fwrite(buf.data(), 1, buf.size(), stdout);
        return;
    }
    static std::vector<std::string> parts;
    parts.resize((size_t)nt);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            format_range(refs, reads, out4, o, reads.nseq * t / nt, reads.nseq * (t + 1) / nt, parts[(size_t)t]);
        });
    for (int t = 0; t < nt; ++t) {
        th[(size_t)t].join();
        fwrite(parts[(size_t)t].data(), 1, parts[(size_t)t].size(), stdout);
    }
}

struct LoadedSketches { std::vector<std::string> names; std::vector<uint64_t> sk; std::vector<int32_t> lens; std::vector<int> ks; int S = 0; std::string policy; };
static bool load_sketch_json(const char* path, LoadedSketches& L);

// The devices of one run (--devices): context 0 builds the reference sketches (rk_set_references on its GPU), the others import
// them (rk_set_reference_sketches: a few MB through the host), all in parallel threads -- the in-process form of the one-rank-per-
// GPU layout of rkmh_amd/cli.py, and the GPU analogue of the reference's -t OpenMP threads (rkmh.cpp:734, :813-898).
// -M: how much of min_num the output needs (rk_set_min_num_bound).  stream / classify print FAIL:DEPTH iff num_mins <= -N
// (rkmh.cpp:938), filter keeps a read iff read_min_lens > 0 (:1292): min(num_mins, bound) answers both, and with it the masked pass
// looks up index keys (and at most `bound` surviving windows per read) in the depth map instead of every window.
// RKMH_EXACT_MIN_NUM=1 keeps the exact form (A/B runs, tests).
static int min_num_bound_for(int compare_with) {
    if (getenv("RKMH_EXACT_MIN_NUM") && atoi(getenv("RKMH_EXACT_MIN_NUM")) != 0) return -1;
    return compare_with < 0 ? 0 : (compare_with >= 0x3fffffff ? -1 : compare_with + 1);
}
// One k-mer size of 17 .. 20 and no --kmer-cache: the enumeration behind the wide k-mer kernel (0.1 s at k = 17 ... 6.7 s at k = 20)
// is kept beside the first reference file, <ref>.k<k>.s<s>.rkkc, so that it is paid once -- the first run at k = 19 / 20 spends it
// (and classifies with the k-mer kernel itself), every later run with these references, k, sketch size and hashing policy loads
// the file in milliseconds (a file for other references is recognised by its tag and rewritten; with --devices the first
// context writes it while it builds its index, the others -- whose indexes are built afterwards -- load it).  RKMH_KMER_CACHE_AUTO=0 / --no-kmer-cache: off (k = 19 / 20 then stay with the hash-space kernel).  Nothing happens
// when the directory cannot be written.
static bool g_no_kmer_cache = false;
static std::string auto_kmer_cache(const Opts& o) {
    if (g_no_kmer_cache || (getenv("RKMH_KMER_CACHE_AUTO") && atoi(getenv("RKMH_KMER_CACHE_AUTO")) == 0)) return "";
    if (o.ks.size() != 1 || o.ks[0] < 17 || o.ks[0] > 20 || o.refs.empty() || !strcmp(o.refs[0], "-")) return "";
    const std::string path = std::string(o.refs[0]) + ".k" + std::to_string(o.ks[0]) + ".s" + std::to_string(o.sketch) + ".rkkc";
    FILE* f = fopen(path.c_str(), "ab"); // (creates it empty when new: an empty file is "no list yet")
    if (!f) return "";
    fclose(f);
    return path;
}
struct DeviceGroup {
    std::vector<rk_ctx*> ctx;
    void create(const Opts& o) {
        std::vector<int> ids = o.devices.empty() ? std::vector<int>{o.device} : o.devices;
        ctx.assign(ids.size(), nullptr);
        std::vector<std::thread> th;
        std::vector<std::string> err(ids.size());
        for (size_t i = 0; i < ids.size(); ++i)
            th.emplace_back([&, i] { if (rk_ctx_create(ids[i], &g_policy, &ctx[i]) != RK_OK) err[i] = rk_last_error(); });
        for (auto& t : th) t.join();
        for (auto& e : err) if (!e.empty()) { fprintf(stderr, "rkmh: %s\n", e.c_str()); exit(1); }
        if (o.kmer_cache && *o.kmer_cache) for (rk_ctx* cx : ctx) CK(rk_set_kmer_cache(cx, o.kmer_cache));
        else { const std::string ac = auto_kmer_cache(o); if (!ac.empty()) for (rk_ctx* cx : ctx) CK(rk_set_kmer_cache(cx, ac.c_str())); }
    }
    // after the references were set on ctx[0]: the same sketches on every other context
    void share_references(const Opts& o) {
        if (ctx.size() < 2) return;
        const int R = rk_num_references(ctx[0]);
        std::vector<uint64_t> sk((size_t)R * (size_t)o.sketch);
        std::vector<int32_t> lens((size_t)R);
        CK(rk_get_reference_sketches(ctx[0], sk.data(), lens.data()));
        std::vector<std::thread> th;
        std::vector<std::string> err(ctx.size());
        for (size_t i = 1; i < ctx.size(); ++i)
            th.emplace_back([&, i] {
                if (rk_set_reference_sketches(ctx[i], sk.data(), lens.data(), R, o.ks.data(), (int)o.ks.size(), o.sketch) != RK_OK) err[i] = rk_last_error();
            });
        for (auto& t : th) t.join();
        for (auto& e : err) if (!e.empty()) { fprintf(stderr, "rkmh: %s\n", e.c_str()); exit(1); }
    }
    void destroy() { for (rk_ctx* c : ctx) rk_ctx_destroy(c); ctx.clear(); }
    size_t size() const { return ctx.size(); }
};
// reads [lo, hi) of a parsed set as a batch of their own (offsets stay absolute: the entry points only use differences and offsets[0])
static inline int64_t share_lo(int64_t n, size_t d, size_t nd) { return n * (int64_t)d / (int64_t)nd; }

// Two passes over ALL reads on several devices (rkmh.cpp:904-948): device d counts and classifies reads [n d / D, n (d+1) / D);
// the depth tables are summed onto device 0 and copied back between the passes, so every device masks with the counts of the WHOLE
// read set, exactly as the reference's threads do with their shared counter.  cnt[0] holds the full table on return.
static void group_run(DeviceGroup& g, const std::function<int(size_t)>& f) {
    const size_t D = g.size();
    std::vector<std::string> err(D);
    std::vector<std::thread> th;
    for (size_t d = 0; d < D; ++d) th.emplace_back([&, d] { if (f(d) != RK_OK) err[d] = rk_last_error(); });
    for (auto& t : th) t.join();
    for (auto& e : err) if (!e.empty()) { fprintf(stderr, "rkmh: %s\n", e.c_str()); exit(1); }
}
// all-reduce of the per-device depth tables inside one process: a binary tree onto device 0 (log2 D rounds, the adds of one round
// on different devices run concurrently) ...
static void sum_counters_on_group(DeviceGroup& g, std::vector<rk_counter*>& cnt) {
    const size_t D = g.size();
    for (size_t step = 1; step < D; step <<= 1)
        group_run(g, [&](size_t d) { return (d % (2 * step) == 0 && d + step < D) ? rk_counter_add(cnt[d], cnt[d + step]) : RK_OK; });
}
// ... and the full table handed back down the same tree
static void share_counters_on_group(DeviceGroup& g, std::vector<rk_counter*>& cnt) {
    const size_t D = g.size();
    size_t top = 1;
    while (top < D) top <<= 1;
    for (size_t step = top >> 1; step >= 1; step >>= 1)
        group_run(g, [&](size_t d) { return (d % (2 * step) == 0 && d + step < D) ? rk_counter_copy(cnt[d + step], cnt[d]) : RK_OK; });
}
// The depth maps of a -M run, one per device: compact (rk_counter_create_compact: only the slots of index keys, a few hundred KB)
// when the output needs min_num only up to bound 0 and every read's hashes fit the sketch, else the reference's full table.
static void make_depth_maps(DeviceGroup& g, uint64_t slots, bool compact, std::vector<rk_counter*>& cnts) {
    for (rk_counter* k : cnts) rk_counter_destroy(k);
    cnts.assign(g.size(), nullptr);
    group_run(g, [&](size_t d) { return compact ? rk_counter_create_compact(g.ctx[d], slots, nullptr, &cnts[d]) : rk_counter_create(g.ctx[d], slots, &cnts[d]); });
}
// RKMH_FULL_DEPTH_MAP=1 keeps the full table (A/B runs, tests)
static bool compact_maps_wanted(int bound, const char* read_map) {
    return bound == 0 && !read_map && !(getenv("RKMH_FULL_DEPTH_MAP") && atoi(getenv("RKMH_FULL_DEPTH_MAP")) != 0);
}
// every read short enough that bottom-s selection cannot matter (conservative: len - k + 1 windows per size) and for the fused kernel
static bool reads_fit_sketch(const rk_seqset& reads, const Opts& o) {
    for (int64_t i = 0; i < reads.nseq; ++i) {
        const int64_t len = (int64_t)(reads.offsets[i + 1] - reads.offsets[i]);
        int64_t nh = 0;
        for (int k : o.ks) nh += len - k + 1 > 0 ? len - k + 1 : 0;
        if (nh > o.sketch || len > 1500) return false;
    }
    return true;
}
static std::atomic<bool> g_need_full{false}; // a count pass met RK_ERR_NEED_FULL: repeat it with full tables

static void two_pass_on_group(DeviceGroup& g, const rk_seqset& reads, uint64_t slots, int min_occ, std::vector<rk_counter*>& cnt,
                              bool pass1, int32_t* out4) {
    const size_t D = g.size();
    if (cnt.empty()) {
        cnt.assign(D, nullptr);
        group_run(g, [&](size_t d) { return rk_counter_create(g.ctx[d], slots, &cnt[d]); });
    }
    if (pass1) {
        group_run(g, [&](size_t d) {
            const int64_t lo = share_lo(reads.nseq, d, D), hi = share_lo(reads.nseq, d + 1, D);
            return rk_count_batch(g.ctx[d], reads.bases, reads.offsets + lo, hi - lo, cnt[d]);
        });
        sum_counters_on_group(g, cnt);
    }
    share_counters_on_group(g, cnt);
    group_run(g, [&](size_t d) {
        const int64_t lo = share_lo(reads.nseq, d, D), hi = share_lo(reads.nseq, d + 1, D);
        int r = rk_set_depth_filter(g.ctx[d], cnt[d], min_occ);
        if (r == RK_OK) r = rk_classify_batch(g.ctx[d], reads.bases, reads.offsets + lo, hi - lo, out4 + lo * 4);
        return r;
    });
}

// ------------------------------------------------------------------------------------------------------------------------
// stream / classify with the FASTQ front end ON THE DEVICE (rk_fastq_slot_*, rkmh_amd/csrc/rk_fastq.hip).  The host no longer
// parses the reads (parse_fastas -> kseq_read, rkmh.cpp:238-263): a coordinator cuts the file into byte ranges of whole records,
// N identical workers each read their range straight into a page-locked buffer, have the GPU split it into records, check it,
// pack it and classify it, and format the lines from the record names where they lie in the raw text; one writer puts the
// blocks back in input order.  Text the device refuses (anything but strictly four lines per record) hands the file over to the
// kseq-grammar scanner from that block on, so the output never depends on which front end ran.
static int granted_cpus_main() {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n < 1) n = (int)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long long per = 0;
        if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
            const long long quota = atoll(q);
            const int c = (int)((quota + per - 1) / per);
            if (c >= 1 && c < n) n = c;
        }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}

// BGZF (bgzip) read files found by raw_eligible: their members are inflated ON THE DEVICE (rk_inflate.hip), thousands per launch,
// by a few workers with device-text slots -- or, RKMH_BGZF_DEVICE=0, by all but two of the CPUs (libdeflate / zlib), job by job
static std::map<std::string, rk_bgzf*> g_bgzf;
static bool bgzf_on_device() {
    static const bool on = [] { const char* e = getenv("RKMH_BGZF_DEVICE"); return !(e && atoi(e) == 0); }();
    return on;
}
static long env_long(const char* name, long dflt, long lo, long hi) { const char* e = getenv(name); if (!e) return dflt; const long v = atol(e); return v < lo || v > hi ? dflt : v; }
static rk_bgzf* bgzf_of(const char* path) { auto it = g_bgzf.find(path); return it == g_bgzf.end() ? nullptr : it->second; }
// ordinary gzip read files (one deflate stream): inflated on the device as well (rk_gunzip.hip), stretch after stretch, one worker per
// file; RKMH_GZIP_DEVICE=0 (or RKMH_BGZF_DEVICE=0) leaves them to zlib and the host scanner
static std::map<std::string, rk_gzip*> g_gzip;
static bool gzip_on_device() {
    static const bool on = [] { const char* e = getenv("RKMH_GZIP_DEVICE"); return !(e && atoi(e) == 0); }();
    return on && bgzf_on_device();
}
static rk_gzip* gzip_of(const char* path) { auto it = g_gzip.find(path); return it == g_gzip.end() ? nullptr : it->second; }

// a regular, uncompressed file that begins with '@' (FASTQ reads) / '>' (FASTA references) -- or, for reads, a BGZF file whose text
// does (*size is then the length of the text); RKMH_BGZF=0 leaves compressed files to the sequential zlib scanner
static bool raw_eligible(const char* path, int64_t* size, char first = '@') {
    if (!path || strcmp(path, "-") == 0) return false;
    if (first == '@' && !(getenv("RKMH_BGZF") && atoi(getenv("RKMH_BGZF")) == 0)) {
        if (rk_bgzf* z = bgzf_of(path)) { *size = (int64_t)rk_bgzf_text_bytes(z); return true; }
        rk_bgzf* z = nullptr;
        if (rk_bgzf_open(path, &z) == RK_OK) {
            if (rk_bgzf_first_byte(z) == '@') { g_bgzf[path] = z; *size = (int64_t)rk_bgzf_text_bytes(z); return true; }
            rk_bgzf_close(z);
            return false;
        }
        if (gzip_on_device()) {
            if (rk_gzip* gz = gzip_of(path)) { *size = (int64_t)rk_gzip_text_bytes_hint(gz); return true; }
            rk_gzip* gz = nullptr;
            if (rk_gzip_open(path, &gz) == RK_OK) {
                if (rk_gzip_first_byte(gz) == '@') { g_gzip[path] = gz; *size = (int64_t)rk_gzip_text_bytes_hint(gz); return true; }
                rk_gzip_close(gz);
                return false;
            }
        }
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    unsigned char magic[2] = {0, 0};
    const bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && pread(fd, magic, 2, 0) >= 1 && magic[0] == (unsigned char)first;
    close(fd);
    if (ok) *size = (int64_t)st.st_size;
    return ok;
}

// Formatted blocks leave in the order of their numbers.  A worker parks its finished block and goes straight on to its next one (it
// only ever waits for memory: at most `window` blocks may be parked ahead of the one due); whoever parks the block that is DUE gives
// the run of consecutive ready blocks their places in the output, in order, and hands them to the writer threads: several of them,
// each with pwrite at the block's final offset, when standard output is a regular file (a file takes ~12 GB/s of buffered writes on the
// test boxes, tools/ubench/file_write.cpp: the output is not what limits the pipeline), one with fwrite otherwise.
struct OrderedOut {
    struct Parked { std::vector<char> buf; size_t len = 0; off_t at = 0; bool keep = false; };
    std::mutex m;
    std::condition_variable cv, cv_task;
    std::map<int64_t, Parked> parked;
    std::deque<Parked> tasks;             // blocks with their place assigned, waiting for a writer
    std::vector<std::vector<char>> spare; // buffers to format the next blocks into
    std::vector<std::thread> writers;
    int64_t next = 0;
    size_t in_flight = 0;                 // tasks queued or being written
    bool assigning = false, closing = false;
    std::atomic<bool> failed{false};      // set by any writer thread
    std::atomic<int64_t> limit{INT64_MAX}; // blocks from this number on are dropped, not written (another front end redoes them)
    bool direct = false;                   // standard output is a regular file not opened for appending
    off_t base = 0, total = 0;
    void lower_limit(int64_t seq) { int64_t cur = limit.load(); while (seq < cur && !limit.compare_exchange_weak(cur, seq)) {} }
    void start(size_t ndev = 1) {
        fflush(stdout);
        struct stat st;
        const int fl = fcntl(1, F_GETFL);
        const bool off_env = getenv("RKMH_OUT_DIRECT") && atoi(getenv("RKMH_OUT_DIRECT")) == 0;
        if (!off_env && fstat(1, &st) == 0 && S_ISREG(st.st_mode) && fl >= 0 && !(fl & O_APPEND)) {
            // ("> out 2>&1": both descriptors are ONE open file; a diagnostic written to stderr during the pass would land at the
            // shared offset, inside the region the blocks are pwritten to -- such a run takes the ordered single-writer path)
            struct stat se;
            const bool same_as_stderr = fstat(2, &se) == 0 && se.st_dev == st.st_dev && se.st_ino == st.st_ino;
            const off_t cur = lseek(1, 0, SEEK_CUR);
            if (cur >= 0 && !same_as_stderr) { direct = true; base = cur; }
        }
        long nw = direct ? 3 : 1; // a pipe or a terminal takes the blocks from ONE thread, in order
        if (direct && ndev > 1) nw = std::min<long>(12, 2 + (long)ndev); // several devices produce lines several times as fast
        if (const char* e = getenv("RKMH_OUT_WRITERS")) { long v = atol(e); if (direct && v >= 1 && v <= 16) nw = v; }
        for (long i = 0; i < nw; ++i)
            writers.emplace_back([this] {
                std::unique_lock<std::mutex> l(m);
                for (;;) {
                    cv_task.wait(l, [&] { return !tasks.empty() || closing; });
                    if (tasks.empty()) return;
                    Parked e = std::move(tasks.front());
                    tasks.pop_front();
                    l.unlock();
                    if (e.keep && e.len) {
                        if (direct) {
                            size_t done_ = 0;
                            while (done_ < e.len) {
                                const ssize_t n = pwrite(1, e.buf.data() + done_, e.len - done_, e.at + (off_t)done_);
                                if (n <= 0) { failed = true; break; }
                                done_ += (size_t)n;
                            }
                        } else if (fwrite(e.buf.data(), 1, e.len, stdout) != e.len) failed = true;
                    }
                    l.lock();
                    if (spare.size() < 32) spare.push_back(std::move(e.buf));
                    --in_flight;
                    cv.notify_all();
                }
            });
    }
    std::vector<char> take_buffer() {
        std::lock_guard<std::mutex> l(m);
        if (spare.empty()) return std::vector<char>();
        std::vector<char> b = std::move(spare.back());
        spare.pop_back();
        return b;
    }
    // buf[0 .. len) are the lines of block seq; the buffer becomes the sink's (a spare one comes back from take_buffer)
    void put(int64_t seq, std::vector<char>&& buf, size_t len, int64_t window) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return seq < next + window && in_flight < (size_t)window; });
        Parked& pk = parked[seq];
        pk.buf = std::move(buf); pk.len = len;
        if (assigning) return; // the thread that is handing blocks out will find this one when its turn comes
        assigning = true;
        for (auto it = parked.find(next); it != parked.end(); it = parked.find(next)) {
            Parked e = std::move(it->second);
            parked.erase(it);
            e.keep = next < limit.load();
            e.at = base + total;
            if (e.keep) total += (off_t)e.len;
            ++next;
            ++in_flight;
            tasks.push_back(std::move(e));
        }
        assigning = false;
        cv_task.notify_all();
        cv.notify_all();
    }
    void finish() {
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return in_flight == 0; });
            closing = true;
        }
        cv_task.notify_all();
        for (auto& t : writers) t.join();
        if (direct && lseek(1, base + total, SEEK_SET) < 0) failed = true; // later output continues behind the blocks
    }
};

// Work handed to a few helper threads: the lines of a device-inflated BGZF job (hundreds of megabytes of text, millions of records)
// are formatted piece by piece by all of them while its worker waits, each piece parked under its own block number
struct FormatPool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> th;
    bool closing = false;
    void start(int n) {
        for (int i = (int)th.size(); i < n; ++i)
            th.emplace_back([this] {
                std::unique_lock<std::mutex> l(m);
                for (;;) {
                    cv.wait(l, [&] { return !q.empty() || closing; });
                    if (q.empty()) return;
                    std::function<void()> f = std::move(q.front());
                    q.pop_front();
                    l.unlock();
                    f();
                    l.lock();
                }
            });
    }
    void run(std::function<void()> f) { { std::lock_guard<std::mutex> l(m); q.push_back(std::move(f)); } cv.notify_one(); }
    void stop() {
        { std::lock_guard<std::mutex> l(m); closing = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
        th.clear();
        closing = false;
    }
};
struct Latch {
    std::mutex m;
    std::condition_variable cv;
    int left = 0;
    void done() { std::lock_guard<std::mutex> l(m); if (--left == 0) cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return left == 0; }); }
};

static const std::vector<const char*>* g_read_paths = nullptr; // the -f files of this run (RawEngine::create: are they all BGZF?)
struct RawEngine {
    // one slot per worker: one block on the device at a time.  (Two slots per worker -- the next block read while the previous one is on
    // the device -- were measured no faster on 64 M reads and 0.2 s slower on 16 M, profiles/r04_e2e_ab.txt, and are gone.)
    struct Worker { rk_fastq_slot* slot = nullptr; size_t dev = 0; bool device_text = false; uint64_t bytes = 0; std::vector<uint8_t> host_text; };
    std::vector<Worker> w;
    uint64_t block = 0;  // text per job: plain files, and BGZF files inflated on the host
    uint64_t mega = 0;   // text per job of BGZF files inflated on the device (0: no such file in this run)
    int pieces = 1;      // block numbers (= output pieces, formatted in parallel) per device-inflated job
    uint64_t gz_stretch = 0; // ordinary gzip files in this run: the most compressed bytes one call takes (0: none)
    bool need_plain_workers = false; // the references go through the workers' page-locked text buffers (refs_through_device)
    FormatPool pool;
    double t_read = 0, t_dev = 0, t_fmt = 0;
    int64_t blocks = 0, records = 0;
    bool create(DeviceGroup& g) {
        if (!w.empty()) return true;
        long mb = 16; // measured (tools/e2e_sweep.py, 16 CPUs): 8-16 MB blocks and 8 workers 82 M reads/s, 32 MB and 14 workers 56-73
        if (const char* e = getenv("RKMH_RAW_BLOCK_KB")) { long v = atol(e); if (v >= 4) { block = (uint64_t)v << 10; mb = 0; } }
        if (mb) block = (uint64_t)mb << 20;
        long nw = std::max(2, granted_cpus_main() * 3 / 8); // 6 of 16 CPUs: 4 / 6 / 8 / 10 workers 77 / 97 / 83 / 95 M reads/s to /dev/null, 53-63 to a file
        // one link is saturated by about six workers (the sweep above: one GPU); with several devices in the process every device
        // gets that many as long as the CPUs last -- not measured (one-GPU boxes), the same reasoning per link
        const long cap = g.size() > 1 ? std::min<long>(64, 6 * (long)g.size()) : 12;
        if (nw > cap) nw = cap;
        const bool dev_inflate = (!g_bgzf.empty() || !g_gzip.empty()) && bgzf_on_device();
        // BGZF inflated on the host: a worker inflates its job's members before the upload (~1 GB/s of text per core with libdeflate,
        // a third of that with zlib) -- the CPUs, not the link, set the rate, so all but two of them work
        if (!g_bgzf.empty() && !dev_inflate) nw = std::max<long>(nw, std::min<long>(32, granted_cpus_main() - 2));
        if (const char* e = getenv("RKMH_RAW_WORKERS")) { long v = atol(e); if (v >= 1 && v <= 64) nw = v; }
        if ((size_t)nw < g.size()) nw = (long)g.size();
        // BGZF inflated on the device: the decode kernel takes the same time for 64 members as for 16 384 (a lane per member, one
        // wave per 64, two waves per CU: up to 32 768 members per launch in one round), so a job is as large as the file allows --
        // a third of the largest file, at most 1 GiB of text -- and three workers per device keep upload, decode, parsing and
        // formatting of consecutive jobs overlapped.  Their slots hold the text on the device only.
        bool all_bgzf = dev_inflate;
        if (dev_inflate) {
            uint64_t largest = 0;
            for (auto& kv : g_bgzf) largest = std::max<uint64_t>(largest, rk_bgzf_text_bytes(kv.second));
            mega = std::min<uint64_t>((uint64_t)1 << 30, std::max<uint64_t>((uint64_t)4 << 20, largest / 3 + ((uint64_t)1 << 20)));
            // (an ordinary gzip file is ONE stream: its stretches follow each other on one worker, so a slot takes a whole file when it can)
            for (auto& kv : g_gzip) mega = std::max<uint64_t>(mega, std::min<uint64_t>((uint64_t)1 << 30, rk_gzip_text_bytes_hint(kv.second) * 5 / 4 + ((uint64_t)8 << 20)));
            if (const long kb = env_long("RKMH_BGZF_JOB_KB", 0, 64, 1536 << 10)) mega = (uint64_t)kb << 10; // (tests: small jobs)
            pieces = (int)std::min<uint64_t>(32, std::max<uint64_t>(1, mega >> 25)); // ~32 MB of text per output piece
            pieces = (int)env_long("RKMH_BGZF_PIECES", pieces, 1, 32);
        }
        const long ndev = dev_inflate ? env_long("RKMH_BGZF_DEVICE_WORKERS", 3, 1, 16) * (long)g.size() : 0;
        // (a run whose read files are ALL BGZF needs no plain-text workers -- their page-locked buffers are the start-up cost of this path)
        if (need_plain_workers || !g_read_paths) all_bgzf = false;
        else for (const char* p : *g_read_paths) if (!g_bgzf.count(p) && !g_gzip.count(p)) all_bgzf = false;
        if (all_bgzf) nw = 0;
        w.resize((size_t)(nw + ndev));
        for (size_t i = 0; i < w.size(); ++i) {
            w[i].dev = i % g.size();
            w[i].device_text = i >= (size_t)nw;
            w[i].bytes = (w[i].device_text ? mega : block) + 64; // (+ 64: a last block of exactly `block` bytes may get its missing newline)
        }
        // each worker creates its own slot when it starts (page-locking ~50 MB takes ~10 ms): the first blocks are on their way
        // while the later workers are still setting up.  Only the first slot is made here, to find out whether the front end works at all.
        if (rk_fastq_slot_create2(g.ctx[0], w[0].bytes, w[0].device_text ? RK_SLOT_DEVICE_TEXT : 0, &w[0].slot) != RK_OK) {
            fprintf(stderr, "rkmh: device FASTQ front end unavailable (%s): using the host scanner\n", rk_last_error());
            w.clear();
            return false;
        }
        if (pieces > 1) pool.start((int)std::min<long>(16, std::max<long>(2, granted_cpus_main() - 2)));
        if (dev_inflate)
            for (auto& kv : g_gzip)
                if (rk_gzip_plan(kv.second, mega) > 0) gz_stretch = std::max<uint64_t>(gz_stretch, rk_gzip_stretch_bytes(kv.second));
        return true;
    }
    // A worker makes its slot when it starts, ONE worker at a time: allocations of several threads queue up inside the runtime anyway,
    // and they slow every other call down while they do (measured: a device-text slot of 841 MB takes 24 ms on its own -- 23 of
    // them page-locking its host arrays --, 60 to 230 ms when three are made at once beside the reference stage, which then takes
    // 0.45 s instead of 0.15).  The first worker's slot exists already (create); the others follow 24 ms apart.
    std::mutex slot_mu;
    void destroy() { pool.stop(); for (auto& x : w) if (x.slot) rk_fastq_slot_destroy(x.slot); w.clear(); }
};
// BGZF files that go to the device: the mapping is page-locked once (14 ms per GB), the DMA engine then reads the compressed
// members out of the page cache itself.  (Refused -- a platform limit -- the uploads go through the runtime's staging.)
static void register_bgzf_mappings() {
    static std::mutex rm;
    static std::map<const rk_bgzf*, bool> registered;
    if (!bgzf_on_device() || (getenv("RKMH_BGZF_REGISTER") && atoi(getenv("RKMH_BGZF_REGISTER")) == 0)) return;
    std::lock_guard<std::mutex> l(rm);
    for (auto& kv : g_bgzf)
        if (!registered.count(kv.second)) registered[kv.second] = rk_host_register_readonly(rk_bgzf_image(kv.second), (size_t)rk_bgzf_file_bytes(kv.second)) == RK_OK;
    static std::map<const rk_gzip*, bool> registered_gz;
    for (auto& kv : g_gzip)
        if (!registered_gz.count(kv.second)) registered_gz[kv.second] = rk_host_register_readonly(rk_gzip_image(kv.second), (size_t)rk_gzip_file_bytes(kv.second)) == RK_OK;
}

// records [lo, hi) of a classified block as a result of their own (the spans index the same text)
static rk_fastq_result sub_result(const rk_fastq_result& r, int64_t lo, int64_t hi) {
    rk_fastq_result p = r;
    p.nrec = hi - lo;
    p.out4 = r.out4 + lo * 4;
    p.name_off = r.name_off + lo; p.name_len = r.name_len + lo;
    p.seq_off = r.seq_off + lo; p.seq_len = r.seq_len + lo; p.qual_off = r.qual_off + lo;
    return p;
}

// the lines of one block (rk_fastq_stream_lines: rk_format.cpp), names taken from where the slot says they lie
static size_t format_raw(const rk_line_parts* lp, const rk_fastq_result& r, const uint8_t* text, std::vector<char>& buf) {
    const size_t need = (size_t)rk_fastq_stream_lines_bound(lp, &r);
    if (buf.size() < need) buf.resize(need + need / 8); // (grows a few times, then stays: no per-block allocation or zero-fill)
    const int64_t n = rk_fastq_stream_lines(lp, &r, text, buf.data(), buf.size());
    if (n < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
    return (size_t)n;
}

// filter's decision: classify_and_count_diff_filter (src/equiv.hpp:324-353): scan from max_shared = prev_best = 0, empty sample name
struct FilterDecision { int ref; int shared; bool diff_ok; };
static FilterDecision filter_decide(const int32_t* r, int min_diff) {
    FilterDecision d;
    if (r[1] <= 0) { d.ref = -1; d.shared = 0; d.diff_ok = 0 > min_diff; return d; }
    d.ref = r[0]; d.shared = r[1];
    const int diff = r[2] - (r[0] == 0 ? 1 : 0); // the stream scan starts at -1, this one at 0
    d.diff_ok = diff > min_diff;
    return d;
}

// filter's output for one block (rk_fastq_filter_records: rk_format.cpp)
static size_t format_filter_raw(const rk_fastq_result& r, const uint8_t* text, const Opts& o, std::vector<char>& buf) {
    const size_t need = (size_t)rk_fastq_filter_records_bound(&r);
    if (buf.size() < need) buf.resize(need + need / 8);
    const int64_t n = rk_fastq_filter_records(&r, text, o.min_matches, o.min_diff, buf.data(), buf.size());
    if (n < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
    return (size_t)n;
}

// what a pass over a file does with each block
enum RawKind { RAW_STREAM, RAW_FILTER, RAW_COUNT };

// A run of read files through the device front end, as ONE pipeline: the workers go from the last blocks of a file straight to the
// first ones of the next (nothing drains between files), the output keeps the order of the command line.  Returns -1 when every
// file was taken whole; else *fail_file (an index into paths) and the byte offset in that file's text (a record start) from which
// the kseq-grammar scanner must continue -- nothing of that file from there on, and nothing of the files behind it, was printed.
// RAW_STREAM prints stream's lines, RAW_FILTER filter's records; RAW_COUNT prints nothing: it is pass 1 of -M (rkmh.cpp:904-910),
// every worker counts its blocks into its device's table cnts[dev] (summed by the caller), and the first refused block ends the pass.
static int64_t stream_files_raw(RawEngine& eng, DeviceGroup& g, const rk_seqset& refs, const Opts& o, const std::vector<const char*>& paths,
                                const std::vector<int64_t>& fsizes, RawKind kind, std::vector<rk_counter*>* cnts, size_t* fail_file) {
    rk_line_parts* lp = nullptr;
    if (kind == RAW_STREAM) CK(rk_line_parts_create(refs.names, refs.name_offsets, refs.nseq, o.sketch, o.min_matches, o.min_diff, &lp));
    const bool counting = kind == RAW_COUNT;
    // bz: compressed (BGZF) -- a job is a run of members [lo, hi); mega: ... inflated on the device: large jobs, device-text slots,
    // eng.pieces block numbers each (else by the worker that takes the job)
    struct File { const char* path = nullptr; int fd = -1; int64_t fsize = 0; rk_bgzf* bz = nullptr; rk_gzip* gz = nullptr; bool gz_own = false, gz_locked = false; bool mega = false; const uint8_t* fmap = nullptr; };
    std::vector<File> files(paths.size());
    const bool want_mmap = getenv("RKMH_RAW_MMAP") && atoi(getenv("RKMH_RAW_MMAP")) != 0;
    for (size_t i = 0; i < paths.size(); ++i) {
        File& F = files[i];
        F.path = paths[i]; F.fsize = fsizes[i];
        F.fd = open(F.path, O_RDONLY);
        if (F.fd < 0) { fprintf(stderr, "rkmh: cannot open %s\n", F.path); fail_exit(); }
        F.bz = bgzf_of(F.path);
        F.gz = F.bz ? nullptr : gzip_of(F.path);
        // (a gzip stream has a position: a file named twice in one run is opened once more for its second turn)
        for (size_t j = 0; F.gz && !F.gz_own && j < i; ++j)
            if (files[j].gz == F.gz) {
                if (rk_gzip_open(F.path, &F.gz) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                F.gz_own = true;
                F.gz_locked = rk_host_register_readonly(rk_gzip_image(F.gz), (size_t)rk_gzip_file_bytes(F.gz)) == RK_OK;
            }
        if (F.gz && eng.mega == 0) { fprintf(stderr, "rkmh: %s: no device-text slots for a gzip stream\n", F.path); fail_exit(); }
        F.mega = (F.bz || F.gz) && eng.mega != 0;
        // RKMH_RAW_MMAP=1: the file is mapped and the mapping page-locked (hipHostRegister): the link reads the page cache itself, the
        // workers copy nothing (tools/ubench/mmap_register.hip)
        if (!F.bz && !F.gz && F.fsize > 0 && want_mmap) {
            void* mp = mmap(nullptr, (size_t)F.fsize, PROT_READ, MAP_SHARED, F.fd, 0);
            if (mp != MAP_FAILED) {
                if (rk_host_register_readonly(mp, (size_t)F.fsize) == RK_OK) F.fmap = (const uint8_t*)mp;
                else munmap(mp, (size_t)F.fsize);
            }
        }
        if (F.mega) register_bgzf_mappings(); // (normally done already, beside the references)
    }
    // file: index into files; at: where the job's first record starts in the (uncompressed) text; ext: its text in the mapped file; nseq: block numbers it owns
    struct Job { size_t file = 0; int64_t seq = 0, lo = 0, hi = 0, at = 0; const uint8_t* ext = nullptr; int64_t nseq = 1; };
    QueueT<Job> jobs_plain, jobs_mega; // (a worker takes the jobs its slot is made for)
    jobs_plain.cap = jobs_mega.cap = eng.w.size();
    bool any_mega = false, any_plain = false, any_gz = false;
    for (const File& F : files) { (F.mega ? any_mega : any_plain) = true; if (F.gz) any_gz = true; }
    OrderedOut out;
    if (!counting) out.start(g.size());
    std::atomic<int64_t> fail_seq{INT64_MAX};
    std::mutex fm;
    std::map<int64_t, std::pair<size_t, int64_t>> fail_at; // block number -> (file, its first byte)
    std::mutex tm;
    std::atomic<int> live_plain{0}, live_mega{0}, needed_mega{INT32_MAX};
    for (auto& x : eng.w) ++(x.device_text ? live_mega : live_plain);
    const int64_t window = (int64_t)eng.w.size() * 4 * (any_mega ? eng.pieces : 1) + 2;
    auto work = [&](size_t wi) {
        RawEngine::Worker& W = eng.w[wi];
        if (!(W.device_text ? any_mega : any_plain)) return; // (no file of this run is for this worker's kind of slot)
        if (W.device_text) { // (... or fewer jobs than workers of it: see needed_mega)
            size_t rank = 0;
            for (size_t j = 0; j < wi; ++j) if (eng.w[j].device_text) ++rank;
            if ((int)rank >= needed_mega.load()) { live_mega.fetch_sub(1); return; }
        }
        QueueT<Job>& jobs = W.device_text ? jobs_mega : jobs_plain;
        std::atomic<int>& live = W.device_text ? live_mega : live_plain;
        const double t_slot = now_s();
        bool slot_ok = true;
        if (!W.slot) {
            std::lock_guard<std::mutex> sl(eng.slot_mu);
            slot_ok = rk_fastq_slot_create2(g.ctx[W.dev], W.bytes, W.device_text ? RK_SLOT_DEVICE_TEXT : 0, &W.slot) == RK_OK;
        }
        // the gunzip work buffers (gigabytes): one worker at a time, and not beside the reference stage (allocations of that size slow
        // every other call of the runtime down while they last: the first slot's, made in create(), cost the references 0.45 s)
        if (slot_ok && W.device_text && eng.gz_stretch && any_gz) {
            std::lock_guard<std::mutex> sl(eng.slot_mu);
            slot_ok = rk_fastq_slot_reserve_gzip(W.slot, eng.gz_stretch) == RK_OK;
        }
        if (!slot_ok) {
            // (memory for another slot ran out: the other workers carry on -- unless this was the last one)
            fprintf(stderr, "rkmh: worker %zu: %s\n", wi, rk_last_error());
            if (live.fetch_sub(1) == 1) { fprintf(stderr, "rkmh: no worker of the device front end could start\n"); fail_exit(); }
            return;
        }
        rk_fastq_slot* const slot = W.slot;
        if (g_timing && W.device_text && now_s() - t_slot > 0.002) fprintf(stderr, "[rkmh timing] worker %zu: device-text slot of %.0f MB made in %.3f s\n", wi, (double)W.bytes / 1e6, now_s() - t_slot);
        if (W.device_text && kind == RAW_FILTER) CK(rk_fastq_slot_set_filter_output(slot, o.min_matches, o.min_diff));
        Job cur;
        double t_rd = 0, t_dv = 0, t_fm = 0;
        int64_t nblk = 0, nrec_ = 0;
        // (block numbers of a job that carry no output of their own)
        auto put_empty = [&](const Job& jb, int64_t from) { if (!counting) for (int64_t e = from; e < jb.nseq; ++e) out.put(jb.seq + e, std::vector<char>(), 0, window); };
        auto declare_failed = [&](const Job& jb) { // the scanner takes the file over from this job's first record
            { std::lock_guard<std::mutex> l(fm); fail_at[jb.seq] = std::make_pair(jb.file, jb.at); }
            if (!counting) out.lower_limit(jb.seq); // (before this block is parked: the sink cannot pass it)
            int64_t curf = fail_seq.load();
            while (jb.seq < curf && !fail_seq.compare_exchange_weak(curf, jb.seq)) {}
        };
        auto finish_block = [&](const Job& jb) {
            const double b = now_s();
            rk_fastq_result res;
            if (rk_fastq_slot_finish(slot, &res) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
            const double c = now_s();
            t_dv += c - b; ++nblk; nrec_ += res.status == 0 ? res.nrec : 0;
            if (res.status != 0) { declare_failed(jb); put_empty(jb, 0); return; }
            const uint8_t* const text = rk_fastq_slot_spans_base(slot);
            auto format_piece = [&](const rk_fastq_result& part, int64_t seq) {
                std::vector<char> buf = out.take_buffer();
                const size_t n = part.nrec == 0 ? 0 : (kind == RAW_FILTER ? format_filter_raw(part, text, o, buf) : format_raw(lp, part, text, buf));
                out.put(seq, std::move(buf), n, window);
            };
            if (jb.nseq == 1) format_piece(res, jb.seq);
            else { // the helpers format the pieces; the slot's arrays stay untouched until all of them are parked
                Latch latch;
                latch.left = (int)jb.nseq;
                for (int64_t e = 0; e < jb.nseq; ++e)
                    eng.pool.run([&, e] {
                        format_piece(sub_result(res, res.nrec * e / jb.nseq, res.nrec * (e + 1) / jb.nseq), jb.seq + e);
                        latch.done();
                    });
                latch.wait();
            }
            t_fm += now_s() - c;
            if (getenv("RKMH_TRACE_JOBS")) fprintf(stderr, "[job] worker %zu parked blocks %lld..%lld (%lld records)\n", wi, (long long)jb.seq, (long long)(jb.seq + jb.nseq - 1), (long long)res.nrec);
        };
        // an ordinary gzip file: ONE job of this worker -- its stretches in order, eng.pieces block numbers each
        auto gzip_file = [&](const Job& fj) {
            const File& F = files[fj.file];
            const int64_t per = eng.pieces, ncalls = fj.nseq / per;
            bool handed_over = false;
            for (int64_t r = 0; r < ncalls; ++r) {
                Job sub; sub.file = fj.file; sub.seq = fj.seq + r * per; sub.nseq = per;
                if (handed_over || sub.seq > fail_seq.load()) { put_empty(sub, 0); continue; }
                const double a = now_s();
                uint64_t nbytes = 0, off = 0;
                const int rc = rk_fastq_slot_load_gzip(slot, F.gz, r, &nbytes, &off);
                if (rc < 0) { fprintf(stderr, "rkmh: %s: %s\n", F.path, rk_last_error()); fail_exit(); }
                sub.at = (int64_t)off;
                if (rc != RK_OK) { // the sequential reader takes the file over from this stretch's first record
                    if (g_timing) fprintf(stderr, "[rkmh timing] %s: the device inflater stops at byte %lld of the text\n", F.path, (long long)off);
                    declare_failed(sub); put_empty(sub, 0); handed_over = true;
                    t_rd += now_s() - a;
                    continue;
                }
                if (nbytes == 0) { put_empty(sub, 0); t_rd += now_s() - a; continue; }
                if (counting) {
                    t_rd += now_s() - a;
                    const double b = now_s();
                    int32_t status = 0; int64_t nrec = 0;
                    const int crc = rk_fastq_slot_count(slot, nbytes, (*cnts)[W.dev], &status, &nrec);
                    if (crc == RK_ERR_NEED_FULL) { g_need_full.store(true); status = 1; }
                    else if (crc != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                    if (status != 0) { declare_failed(sub); handed_over = true; }
                    t_dv += now_s() - b; ++nblk; nrec_ += nrec;
                    continue;
                }
                if (rk_fastq_slot_submit(slot, nbytes) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                t_rd += now_s() - a;
                finish_block(sub);
                if (fail_seq.load() <= sub.seq) handed_over = true; // (text that is not four lines per record)
            }
            // (the file's 4 MB on the device stay until the run ends: hipFree waits for every stream of the device to drain -- seconds, while
            // the other workers' kernels run, tools/ubench/malloc_vs_kernels.hip -- and holds the runtime's lock meanwhile)
        };
        for (;;) {
            if (!jobs.pop(&cur)) break;
            // (a failure is declared at the first block number of the failing worker's own job: never inside another job's run)
            if (cur.seq > fail_seq.load()) { put_empty(cur, 0); continue; } // the scanner will redo this range
            if (files[cur.file].gz) { gzip_file(cur); continue; }
            const File& F = files[cur.file];
            rk_bgzf* const bz = F.bz;
            const char* const path = F.path;
            const int fd = F.fd;
            const int64_t fsize = F.fsize;
            const uint8_t* const fmap = F.fmap;
            const double a = now_s();
            uint64_t nbytes = 0;
            bool refused = false; // (BGZF: text that does not begin with '@', or a job whose records outgrow the slot)
            if (bz) {
                uint64_t off = 0;
                // the members inflated on the device (which may hand a job back: a member it cannot decode, a failed CRC-32 -- the host
                // inflater then reports the damage) or by this thread
                int rc = W.device_text ? rk_fastq_slot_load_bgzf(slot, bz, cur.lo, cur.hi, &nbytes, &off) : 1;
                if (rc < 0) { fprintf(stderr, "rkmh: %s: %s\n", path, rk_last_error()); fail_exit(); }
                if (rc != RK_OK) {
                    uint8_t* text = rk_fastq_slot_text(slot);
                    if (W.device_text) { if (W.host_text.size() < W.bytes) W.host_text.resize(W.bytes); text = W.host_text.data(); }
                    rc = rk_bgzf_fastq_records(bz, cur.lo, cur.hi, text, W.bytes - 1, &nbytes, &off);
                    if (rc == 1 || rc == RK_ERR_LIMIT) { refused = true; nbytes = 0; }
                    else if (rc != RK_OK) { fprintf(stderr, "rkmh: %s: %s\n", path, rk_last_error()); fail_exit(); }
                    if (!refused && cur.hi == rk_bgzf_members(bz) && nbytes && text[nbytes - 1] != '\n') text[nbytes++] = '\n';
                    if (!refused && W.device_text && rk_fastq_slot_set_source(slot, text) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                }
                cur.at = (int64_t)off;
            } else if (fmap && !(cur.hi == fsize && fmap[fsize - 1] != '\n')) { // (a last block without its newline is copied, to get one)
                cur.at = cur.lo;
                cur.ext = fmap + cur.lo;
                nbytes = (uint64_t)(cur.hi - cur.lo);
                if (rk_fastq_slot_set_source(slot, cur.ext) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
            } else {
                uint8_t* text = rk_fastq_slot_text(slot);
                cur.at = cur.lo;
                int64_t have = 0;
                while (have < cur.hi - cur.lo) {
                    const ssize_t n = pread(fd, text + have, (size_t)(cur.hi - cur.lo - have), (off_t)(cur.lo + have));
                    if (n <= 0) { fprintf(stderr, "rkmh: read error on %s\n", path); fail_exit(); } // (the other workers may be waiting for this block)
                    have += n;
                }
                nbytes = (uint64_t)(cur.hi - cur.lo);
                if (cur.hi == fsize && nbytes && text[nbytes - 1] != '\n') text[nbytes++] = '\n'; // a last line without its newline (the slot holds 64 spare bytes)
            }
            if (refused) {
                declare_failed(cur);
                put_empty(cur, 0);
                t_rd += now_s() - a;
                continue;
            }
            if (counting) {
                t_rd += now_s() - a;
                const double b = now_s();
                int32_t status = 0; int64_t nrec = 0;
                const int crc = rk_fastq_slot_count(slot, nbytes, (*cnts)[W.dev], &status, &nrec);
                if (crc == RK_ERR_NEED_FULL) { g_need_full.store(true); status = 1; } // a read with more hashes than the sketch keeps: the pass ends, the caller repeats it with full tables
                else if (crc != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                if (status != 0) declare_failed(cur);
                t_dv += now_s() - b; ++nblk; nrec_ += nrec;
                continue;
            }
            if (rk_fastq_slot_submit(slot, nbytes) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
            t_rd += now_s() - a;
            finish_block(cur);
        }
        std::lock_guard<std::mutex> l(tm);
        eng.t_read += t_rd; eng.t_dev += t_dv; eng.t_fmt += t_fm; eng.blocks += nblk; eng.records += nrec_;
    };
    // how many jobs the device-text workers will share (an ordinary gzip file is one job, a BGZF file a few): a worker without a job
    // to expect does not start -- its slot and work buffers are gigabytes of allocations that slow the others down while they are made
    // (one gzip file: 0.36 s as a command with one such worker setting up, 0.6 - 0.77 s with four or five)
    size_t mega_jobs = 0;
    for (const File& F : files) {
        if (F.gz) ++mega_jobs;
        else if (F.bz && F.mega) {
            const uint64_t target = eng.mega > ((uint64_t)1 << 20) ? eng.mega - ((uint64_t)1 << 18) : eng.mega * 3 / 4;
            std::vector<int64_t> first((size_t)rk_bgzf_members(F.bz) + 4);
            const int64_t nj = rk_bgzf_plan_members(F.bz, target, 16381, first.data(), (int64_t)first.size());
            mega_jobs += nj > 0 ? (size_t)nj : 1;
        }
    }
    needed_mega.store((int)std::min<size_t>(mega_jobs, (size_t)INT32_MAX));
    std::vector<std::thread> workers;
    for (size_t i = 0; i < eng.w.size(); ++i) workers.emplace_back(work, i);
    // coordinator: ranges of whole records, file after file.  The end of a range is the last record start (four-line rule,
    // rk_fastq_cut) inside a window in front of its nominal end; a range that is cut wrongly (possible only in text that is not
    // four lines per record) is refused by the device and the scanner takes over from its first byte.
    int64_t seq = 0;
    for (size_t fi = 0; fi < files.size() && fail_seq.load() == INT64_MAX; ++fi) {
        const File& F = files[fi];
        if (F.gz) { // one job: the file's stretches, in order, on one worker
            const int64_t ncalls = rk_gzip_plan(F.gz, eng.mega);
            if (ncalls < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
            Job jb; jb.file = fi; jb.seq = seq; jb.nseq = ncalls * eng.pieces;
            seq += jb.nseq;
            jobs_mega.push(jb);
            continue;
        }
        if (F.bz) { // jobs = runs of members holding about a block of text (the records are cut after inflating)
            const uint64_t per_job = F.mega ? eng.mega : eng.block;
            const uint64_t target = per_job > ((uint64_t)1 << 20) ? per_job - ((uint64_t)1 << 18) : per_job * 3 / 4;
            std::vector<int64_t> first((size_t)rk_bgzf_members(F.bz) + 4);
            // (device jobs: at most 16 381 members + the three around them = 256 waves of 64 members: two launches fill the chip exactly)
            const int64_t nj = rk_bgzf_plan_members(F.bz, target, F.mega ? 16381 : INT64_MAX, first.data(), (int64_t)first.size());
            if (nj < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
            const int64_t per = F.mega ? eng.pieces : 1;
            for (int64_t j = 0; j < nj && fail_seq.load() == INT64_MAX; ++j) {
                Job jb; jb.file = fi; jb.seq = seq; jb.nseq = per; jb.lo = first[(size_t)j]; jb.hi = first[(size_t)j + 1];
                seq += per;
                (F.mega ? jobs_mega : jobs_plain).push(jb);
            }
            continue;
        }
        std::vector<uint8_t> win;
        int64_t pos = 0;
        const int64_t B = (int64_t)eng.block, fsize = F.fsize;
        while (pos < fsize && fail_seq.load() == INT64_MAX) {
            int64_t hi = fsize;
            if (fsize - pos > B) {
                int64_t wlen = 1 << 16;
                hi = -1;
                while (hi < 0) {
                    if (wlen > B - 1) wlen = B - 1;
                    const int64_t wlo = pos + B - wlen; // the window ends at the nominal end of the range
                    win.resize((size_t)wlen);
                    int64_t got = 0;
                    while (got < wlen) {
                        const ssize_t n = pread(F.fd, win.data() + got, (size_t)(wlen - got), (off_t)(wlo + got));
                        if (n <= 0) break;
                        got += n;
                    }
                    const int64_t cut = got == wlen ? rk_fastq_cut(win.data(), (uint64_t)wlen) : -1;
                    if (cut > 0) hi = wlo + cut;
                    else if (wlen >= B - 1) break; // no record start anywhere in the range: not for the device
                    else wlen *= 8;
                }
                if (hi < 0) { // hand the file over from here
                    std::lock_guard<std::mutex> l(fm);
                    fail_at[seq] = std::make_pair(fi, pos);
                    if (!counting) out.lower_limit(seq);
                    int64_t cur = fail_seq.load();
                    while (seq < cur && !fail_seq.compare_exchange_weak(cur, seq)) {}
                    break;
                }
            }
            Job j; j.file = fi; j.seq = seq++; j.lo = pos; j.hi = hi;
            jobs_plain.push(j);
            pos = hi;
        }
    }
    jobs_plain.finish();
    jobs_mega.finish();
    for (auto& t : workers) t.join();
    if (!counting) out.finish();
    rk_line_parts_destroy(lp);
    for (File& F : files) {
        if (F.fmap) { rk_host_unregister(F.fmap); munmap((void*)F.fmap, (size_t)F.fsize); }
        if (F.gz_own) { if (F.gz_locked) rk_host_unregister(rk_gzip_image(F.gz)); rk_gzip_close(F.gz); }
        close(F.fd);
    }
    if (out.failed) { fprintf(stderr, "rkmh: write error on standard output\n"); fail_exit(); }
    const int64_t fs = fail_seq.load();
    if (fs == INT64_MAX) return -1;
    if (fail_file) *fail_file = fail_at[fs].first;
    return fail_at[fs].second;
}


// -M with the device front end (rkmh.cpp:904-948 without holding the reads in RAM): pass 1 counts every file's blocks, the depth
// tables are summed over the devices and become every context's mask, pass 2 reads the files again and prints.  false: some block
// is not four lines per record -- nothing was printed, the tables are clear again and the caller takes the parse-everything path.
static bool two_pass_raw(RawEngine& eng, DeviceGroup& g, const rk_seqset& refs, const Opts& o, const std::vector<int64_t>& sizes,
                         std::vector<rk_counter*>& cnts, RawKind kind, double& t0, uint64_t slots) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t ff = 0;
        if (stream_files_raw(eng, g, refs, o, o.reads, sizes, RAW_COUNT, &cnts, &ff) < 0) break;
        if (attempt == 0 && g_need_full.exchange(false) && rk_counter_is_compact(cnts[0])) {
            // a compact depth map cannot serve reads whose hashes exceed the sketch: the same pass again into full tables
            if (g_timing) fprintf(stderr, "[rkmh timing] %s: reads with more hashes than the sketch keeps: pass 1 restarts with full depth tables\n", o.reads[ff]);
            make_depth_maps(g, slots, false, cnts);
            continue;
        }
        group_run(g, [&](size_t d) { return rk_counter_clear(cnts[d]); });
        if (g_timing) fprintf(stderr, "[rkmh timing] %s: not four lines per record: the host scanner reads the run\n", o.reads[ff]);
        return false;
    }
    tick("pass 1 (device front end + count)", t0);
    sum_counters_on_group(g, cnts);
    share_counters_on_group(g, cnts);
    group_run(g, [&](size_t d) { return rk_set_depth_filter(g.ctx[d], cnts[d], o.min_occ); });
    tick("depth tables summed, mask built", t0);
    {
        size_t ff = 0;
        if (stream_files_raw(eng, g, refs, o, o.reads, sizes, kind, nullptr, &ff) >= 0) {
            fprintf(stderr, "rkmh: %s changed between the two passes\n", o.reads[ff]);
            fail_exit();
        }
    }
    fflush(stdout);
    tick("pass 2 (device front end + classify)", t0);
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// Packed reads: `rkmh pack` writes them, `stream|filter -F` reads them (include/rkmh_amd.h, "PACKED READS").  The reference parses
// -F/--pre-reads and does nothing with it (src/rkmh.cpp:659-664); here it names reads that were parsed ONCE: 2 bits per base, the
// names and (optionally) the quality strings kept for the host -- a run then moves ~42 bytes per 150-base read over the link
// instead of 315 of FASTQ text, parses nothing, and formats its lines from the names where they lie in the mapped file.
static void help_pack() {
    fprintf(stderr,
            "rkmh pack -f <reads.fq|fa[.gz]> [-f ...] -o <out.rkp> [--no-quals] [--block-reads <n>]\n"
            "  writes the reads as a packed file (2 bits per base, names, quality strings unless --no-quals) that\n"
            "  `rkmh stream|filter -F <out.rkp>` classifies without parsing; independent of k, sketch size and hashing policy\n");
}
static int main_pack(int argc, char** argv) {
    std::vector<const char*> files;
    const char* outp = nullptr;
    bool keep_quals = true;
    long block_reads = 1 << 20;
    if (argc <= 2) { help_pack(); exit(1); }
    static struct option long_options[] = {{"help", no_argument, 0, 'h'}, {"fasta", required_argument, 0, 'f'}, {"output", required_argument, 0, 'o'},
                                           {"no-quals", no_argument, 0, 1010}, {"block-reads", required_argument, 0, 1011}, {"threads", required_argument, 0, 't'}, {0, 0, 0, 0}};
    optind = 2;
    int c;
    while ((c = getopt_long(argc, argv, "hf:o:t:", long_options, nullptr)) != -1) {
        switch (c) {
            case 'f': files.push_back(optarg); break;
            case 'o': outp = optarg; break;
            case 't': break;
            case 1010: keep_quals = false; break;
            case 1011: block_reads = atol(optarg); break;
            default: help_pack(); exit(1);
        }
    }
    if (files.empty() || !outp) { help_pack(); exit(1); }
    if (block_reads < 1024 || block_reads > (16 << 20)) { fprintf(stderr, "rkmh pack: --block-reads must lie between 1024 and 16777216\n"); exit(1); }
    FILE* fo = fopen(outp, "wb");
    if (!fo) { fprintf(stderr, "rkmh pack: cannot write %s\n", outp); exit(1); }
    rk_packed_header hdr;
    memset(&hdr, 0, sizeof hdr);
    memcpy(hdr.magic, RK_PACKED_MAGIC, 8);
    hdr.version = 1;
    std::vector<rk_packed_block> dir;
    uint64_t at = 0;
    bool quals_everywhere = keep_quals;
    auto put = [&](const void* p, size_t n) { if (n && fwrite(p, 1, n, fo) != n) { fprintf(stderr, "rkmh pack: write error on %s\n", outp); exit(1); } at += n; };
    auto align16 = [&]() { static const char z[16] = {0}; const size_t pad = (size_t)((16 - (at & 15)) & 15); put(z, pad); };
    put(&hdr, sizeof hdr); // (rewritten at the end)
    const int nt = std::max(1, std::min(granted_cpus_main(), 16));
    std::vector<uint8_t> b2;
    std::vector<std::vector<rk_packed_exception>> exc_t((size_t)nt);
    std::vector<uint32_t> offs, noffs;
    std::vector<char> names;
    for (const char* path : files) {
        rk_reader* rd = nullptr;
        CK(rk_reader_open(path, &rd));
        if (!keep_quals) rk_reader_set_options(rd, RK_READER_NO_QUALS);
        for (;;) {
            rk_seqset s;
            CK(rk_reader_next(rd, block_reads, (uint64_t)3 << 30, &s));
            if (s.nseq == 0) { rk_seqset_free(&s); break; }
            const uint64_t b0 = s.offsets[0], nb = s.offsets[s.nseq] - b0;
            if (nb >= ((uint64_t)1 << 32) - 64 || s.nseq > 0x7ffffff0ll) { fprintf(stderr, "rkmh pack: a block of more than 4 G bases\n"); exit(1); }
            rk_packed_block blk;
            memset(&blk, 0, sizeof blk);
            blk.nrec = (uint32_t)s.nseq; blk.nbases = nb;
            offs.resize((size_t)s.nseq + 1);
            uint32_t maxlen = 0;
            for (int64_t i = 0; i <= s.nseq; ++i) offs[(size_t)i] = (uint32_t)(s.offsets[i] - b0);
            for (int64_t i = 0; i < s.nseq; ++i) maxlen = std::max(maxlen, offs[(size_t)i + 1] - offs[(size_t)i]);
            blk.max_len = maxlen;
            // 2-bit bases and exceptions: pieces of whole bytes (4 bases), a thread each
            b2.assign((size_t)((nb + 3) / 4), 0);
            {
                std::vector<std::thread> th;
                const uint64_t per = (((nb + (uint64_t)nt - 1) / (uint64_t)nt) + 3) & ~(uint64_t)3;
                for (int t = 0; t < nt; ++t)
                    th.emplace_back([&, t] {
                        const uint64_t lo = std::min(nb, per * (uint64_t)t), hi = std::min(nb, lo + per);
                        auto& ex = exc_t[(size_t)t];
                        ex.resize((size_t)(hi - lo) + 1);
                        const int64_t ne = rk_packed_encode(s.bases + b0 + lo, hi - lo, lo, b2.data() + lo / 4, ex.data(), ex.size());
                        if (ne < 0) { fprintf(stderr, "rkmh pack: %s\n", rk_last_error()); fail_exit(); }
                        ex.resize((size_t)ne);
                    });
                for (auto& t : th) t.join();
            }
            noffs.resize((size_t)s.nseq + 1);
            names.clear();
            for (int64_t i = 0; i < s.nseq; ++i) {
                noffs[(size_t)i] = (uint32_t)names.size();
                const char* nm = s.names + s.name_offsets[i];
                names.insert(names.end(), nm, nm + (s.name_offsets[i + 1] - s.name_offsets[i] - 1)); // (the offsets include the NUL)
            }
            noffs[(size_t)s.nseq] = (uint32_t)names.size();
            if (names.size() >= ((uint64_t)1 << 32)) { fprintf(stderr, "rkmh pack: more than 4 GB of names in one block\n"); exit(1); }
            blk.name_bytes = names.size();
            align16(); blk.offsets_off = at; put(offs.data(), offs.size() * 4);
            align16(); blk.bases_off = at; put(b2.data(), b2.size());
            { static const char z[16] = {0}; put(z, 16); } // (the bases are uploaded in whole dwords; the unpacked tail is never read)
            align16(); blk.exc_off = at;
            uint64_t nexc = 0;
            for (auto& ex : exc_t) { put(ex.data(), ex.size() * sizeof(rk_packed_exception)); nexc += ex.size(); }
            if (nexc > 0xffffffffull) { fprintf(stderr, "rkmh pack: too many non-ACGT bases in one block\n"); exit(1); }
            blk.nexc = (uint32_t)nexc;
            align16(); blk.name_offsets_off = at; put(noffs.data(), noffs.size() * 4);
            align16(); blk.names_off = at; put(names.data(), names.size());
            { static const char z[32] = {0}; put(z, 32); } // (the formatters copy names in 16-byte steps)
            if (keep_quals && s.quals) { align16(); blk.quals_off = at; put(s.quals + b0, (size_t)nb); }
            else quals_everywhere = false;
            dir.push_back(blk);
            hdr.nreads += (uint64_t)s.nseq; hdr.nbases += nb;
            rk_seqset_free(&s);
        }
        rk_reader_close(rd);
    }
    if (!quals_everywhere) for (auto& b : dir) b.quals_off = 0; // (all or nothing: a file that keeps qualities keeps them for every read)
    align16();
    hdr.directory_off = at; hdr.nblocks = dir.size(); hdr.flags = quals_everywhere && !dir.empty() ? RK_PACKED_QUALS : 0u;
    put(dir.data(), dir.size() * sizeof(rk_packed_block));
    { static const char z[64] = {0}; put(z, 64); }
    if (fseek(fo, 0, SEEK_SET) != 0 || fwrite(&hdr, sizeof hdr, 1, fo) != 1 || fclose(fo) != 0) { fprintf(stderr, "rkmh pack: write error on %s\n", outp); exit(1); }
    fprintf(stderr, "rkmh pack: %llu reads, %llu bases in %zu blocks%s -> %s (%.1f bytes per read)\n", (unsigned long long)hdr.nreads, (unsigned long long)hdr.nbases, dir.size(),
            hdr.flags & RK_PACKED_QUALS ? ", with qualities" : "", outp, hdr.nreads ? (double)(at + dir.size() * sizeof(rk_packed_block)) / (double)hdr.nreads : 0.0);
    return 0;
}

struct PackedFile {
    const char* path = nullptr;
    const uint8_t* map = nullptr;
    size_t size = 0;
    const rk_packed_header* hdr = nullptr;
    const rk_packed_block* dir = nullptr;
};
// maps and checks a packed file (every section inside the file, counts consistent); exits with a message otherwise
static PackedFile packed_open(const char* path) {
    PackedFile pf;
    pf.path = path;
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || st.st_size < (off_t)sizeof(rk_packed_header)) { fprintf(stderr, "rkmh: cannot read packed reads from %s\n", path); exit(1); }
    void* mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (mp == MAP_FAILED) { fprintf(stderr, "rkmh: cannot map %s\n", path); exit(1); }
    pf.map = (const uint8_t*)mp; pf.size = (size_t)st.st_size;
    pf.hdr = reinterpret_cast<const rk_packed_header*>(pf.map);
    auto bad = [&](const char* why) { fprintf(stderr, "rkmh: %s is not a packed read file of this build (%s): write it with `rkmh pack`\n", path, why); exit(1); };
    if (memcmp(pf.hdr->magic, RK_PACKED_MAGIC, 8) != 0 || pf.hdr->version != 1) bad("magic / version");
    const uint64_t nb = pf.hdr->nblocks, doff = pf.hdr->directory_off;
    if ((doff & 15) || doff > pf.size || nb > (pf.size - doff) / sizeof(rk_packed_block)) bad("directory");
    pf.dir = reinterpret_cast<const rk_packed_block*>(pf.map + doff);
    uint64_t nreads = 0;
    for (uint64_t i = 0; i < nb; ++i) {
        const rk_packed_block& b = pf.dir[i];
        auto inside = [&](uint64_t off, uint64_t n) { return (off & 15) == 0 && off <= doff && n <= doff - off; };
        if (!inside(b.offsets_off, ((uint64_t)b.nrec + 1) * 4) || !inside(b.bases_off, (b.nbases + 3) / 4 + 16) || !inside(b.exc_off, (uint64_t)b.nexc * 8) ||
            !inside(b.name_offsets_off, ((uint64_t)b.nrec + 1) * 4) || !inside(b.names_off, b.name_bytes + 32) || (b.quals_off && !inside(b.quals_off, b.nbases)))
            bad("a block's sections");
        const uint32_t* so = reinterpret_cast<const uint32_t*>(pf.map + b.offsets_off);
        const uint32_t* no = reinterpret_cast<const uint32_t*>(pf.map + b.name_offsets_off);
        if (so[0] != 0 || so[b.nrec] != b.nbases || no[0] != 0 || no[b.nrec] != b.name_bytes) bad("a block's offsets");
        nreads += b.nrec;
    }
    if (nreads != pf.hdr->nreads) bad("read count");
    return pf;
}

// The blocks of the packed files through the devices: per device a few workers, each with a packed slot (rk_packed_slot_*): upload the
// block's offsets, 2-bit bases and exceptions from the mapping, classify (or count: pass 1 of -M), format the lines from the names in
// the mapping -- large blocks in pieces, by the helper threads -- and park them in input order.
static void stream_packed(DeviceGroup& g, const rk_seqset& refs, const Opts& o, const std::vector<PackedFile>& files, RawKind kind, std::vector<rk_counter*>* cnts) {
    const bool counting = kind == RAW_COUNT;
    rk_line_parts* lp = nullptr;
    if (kind == RAW_STREAM) CK(rk_line_parts_create(refs.names, refs.name_offsets, refs.nseq, o.sketch, o.min_matches, o.min_diff, &lp));
    struct Job { size_t file; uint64_t block; int64_t seq, nseq; };
    std::vector<Job> jobs;
    uint64_t max_reads = 1, max_bases = 16;
    const int64_t PIECE = 1 << 18; // reads per output piece
    int64_t seq = 0;
    for (size_t f = 0; f < files.size(); ++f)
        for (uint64_t b = 0; b < files[f].hdr->nblocks; ++b) {
            const rk_packed_block& blk = files[f].dir[b];
            max_reads = std::max<uint64_t>(max_reads, blk.nrec); max_bases = std::max<uint64_t>(max_bases, blk.nbases);
            const int64_t pieces = std::max<int64_t>(1, ((int64_t)blk.nrec + PIECE - 1) / PIECE);
            jobs.push_back(Job{f, b, seq, pieces});
            seq += pieces;
        }
    static std::map<const uint8_t*, bool> registered; // (a mapping is page-locked once; both passes of -M use it)
    for (const PackedFile& pf : files)
        if (!registered.count(pf.map) && !(getenv("RKMH_PACKED_REGISTER") && atoi(getenv("RKMH_PACKED_REGISTER")) == 0)) {
            const double a = now_s();
            registered[pf.map] = rk_host_register_readonly(pf.map, pf.size) == RK_OK;
            if (g_timing) fprintf(stderr, "[rkmh timing] %s: mapping of %.0f MB %s in %.3f s\n", pf.path, (double)pf.size / 1e6, registered[pf.map] ? "page-locked" : "NOT page-locked (uploads are staged by the runtime)", now_s() - a);
        }
    const size_t nw = (size_t)env_long("RKMH_PACKED_WORKERS", 3, 1, 16) * g.size();
    OrderedOut out;
    if (!counting) out.start(g.size());
    FormatPool pool;
    if (!counting) pool.start((int)std::min<long>(16, std::max<long>(2, granted_cpus_main() - 2)));
    int64_t most_pieces = 1;
    for (const Job& jb : jobs) most_pieces = std::max(most_pieces, jb.nseq);
    // every worker has two slots: while the pool formats the lines of one block (from the rows in that slot's page-locked buffer) the
    // worker's next block is on the device in the other.  At most 2 nw consecutive blocks are open at a time, so a piece never waits
    // in put() for a piece that is queued behind it
    const int64_t window = (int64_t)nw * 2 * most_pieces + 2;
    std::atomic<size_t> next{0};
    std::mutex tm;
    double t_dev = 0, t_fmt = 0;
    auto work = [&](size_t wi) {
        rk_packed_slot* slot[2] = {nullptr, nullptr};
        rk_fastq_result res[2];
        Latch latch[2];
        const size_t dev = wi % g.size();
        const int nslot = counting ? 1 : 2;
        for (int i = 0; i < nslot; ++i)
            if (rk_packed_slot_create(g.ctx[dev], max_reads, max_bases, &slot[i]) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
        double dv = 0, fm = 0;
        int cur = 0;
        for (size_t j = next.fetch_add(1); j < jobs.size(); j = next.fetch_add(1)) {
            const Job& jb = jobs[j];
            const PackedFile& pf = files[jb.file];
            const rk_packed_block& blk = pf.dir[jb.block];
            const double a = now_s();
            if (counting) {
                const int rc = rk_packed_slot_count(slot[0], &blk, pf.map, (*cnts)[dev]);
                if (rc == RK_ERR_NEED_FULL) g_need_full.store(true);
                else if (rc != RK_OK) { fprintf(stderr, "rkmh: %s: %s\n", pf.path, rk_last_error()); fail_exit(); }
                dv += now_s() - a;
                continue;
            }
            latch[cur].wait(); // the lines of the block this slot held before are with the sink
            const double a2 = now_s();
            if (rk_packed_slot_classify(slot[cur], &blk, pf.map, &res[cur]) != RK_OK) { fprintf(stderr, "rkmh: %s: %s\n", pf.path, rk_last_error()); fail_exit(); }
            const double b = now_s();
            const rk_fastq_result* const rs = &res[cur];
            Latch* const lt = &latch[cur];
            { std::lock_guard<std::mutex> l(lt->m); lt->left = (int)jb.nseq; }
            for (int64_t e = 0; e < jb.nseq; ++e)
                pool.run([&out, &o, &pf, &blk, &jb, rs, lt, lp, kind, window, e] {
                    const int64_t lo = rs->nrec * e / jb.nseq, hi = rs->nrec * (e + 1) / jb.nseq;
                    std::vector<char> buf = out.take_buffer();
                    size_t n = 0;
                    if (hi > lo) {
                        const rk_fastq_result part = sub_result(*rs, lo, hi);
                        if (kind == RAW_FILTER) {
                            const size_t need = (size_t)rk_packed_filter_records_bound(&part);
                            if (buf.size() < need) buf.resize(need + need / 8);
                            const int64_t w = rk_packed_filter_records(&part, &blk, pf.map, o.min_matches, o.min_diff, buf.data(), buf.size());
                            if (w < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
                            n = (size_t)w;
                        } else n = format_raw(lp, part, pf.map + blk.names_off, buf);
                    }
                    out.put(jb.seq + e, std::move(buf), n, window);
                    lt->done();
                });
            dv += b - a2; fm += a2 - a;
            cur ^= 1;
        }
        for (int i = 0; i < nslot; ++i) { latch[i].wait(); rk_packed_slot_destroy(slot[i]); }
        std::lock_guard<std::mutex> l(tm);
        t_dev += dv; t_fmt += fm;
    };
    std::vector<std::thread> th;
    for (size_t i = 0; i < nw; ++i) th.emplace_back(work, i);
    for (auto& t : th) t.join();
    pool.stop();
    if (!counting) out.finish();
    rk_line_parts_destroy(lp);
    if (out.failed) { fprintf(stderr, "rkmh: write error on standard output\n"); fail_exit(); }
    if (g_timing) fprintf(stderr, "[rkmh timing] packed reads: %zu blocks; upload + classify %.3f s, waiting for the lines of an earlier block %.3f s (summed over %zu workers)\n", jobs.size(), t_dev, t_fmt, nw);
}

// stream / filter over packed files, with or without -M (two passes: count, sum over the devices, mask, classify)
static void run_packed(DeviceGroup& g, const rk_seqset& refs, const Opts& o, const std::vector<const char*>& paths, RawKind kind, uint64_t slots, int bound, double& t0) {
    std::vector<PackedFile> files;
    for (const char* p : paths) files.push_back(packed_open(p));
    if (kind == RAW_FILTER && !files.empty() && !(files[0].hdr->flags & RK_PACKED_QUALS) && g_timing) fprintf(stderr, "[rkmh timing] %s keeps no qualities: filter prints empty quality lines\n", files[0].path);
    if (o.read_depth) {
        std::vector<rk_counter*> cnts;
        bool compact = compact_maps_wanted(bound, nullptr);
        for (int attempt = 0; attempt < 2; ++attempt) {
            make_depth_maps(g, slots, compact, cnts);
            g_need_full.store(false);
            stream_packed(g, refs, o, files, RAW_COUNT, &cnts);
            if (!g_need_full.exchange(false)) break;
            if (!compact) { fprintf(stderr, "rkmh: the count pass failed\n"); fail_exit(); }
            compact = false; // a read with more hashes than the sketch keeps: the pass again into full tables
        }
        tick("pass 1 (packed reads, count)", t0);
        sum_counters_on_group(g, cnts);
        share_counters_on_group(g, cnts);
        group_run(g, [&](size_t d) { return rk_set_depth_filter(g.ctx[d], cnts[d], o.min_occ); });
        tick("depth tables summed, mask built", t0);
        stream_packed(g, refs, o, files, kind, nullptr);
        tick("pass 2 (packed reads, classify)", t0);
        return;
    }
    stream_packed(g, refs, o, files, kind, nullptr);
    tick("packed reads: classify + format", t0);
}

// The -r files through the device (rk_fasta_load_*, rkmh_amd/csrc/rk_fasta.hip) instead of parse_fastas (rkmh.cpp:238-263): the
// workers of the read pipeline pread the raw text into their page-locked buffers and upload it, the GPU strips header lines and
// line ends, and the references are sketched from the packed bases where they lie -- the host never sees a base.  Worth its set-up
// for genome-sized references (BASELINE config 4: 3.1 GB of FASTA, where the host parser was the longest stage of the run);
// RKMH_RAW_REFS=1 forces it for any size, =0 turns it off.  false: not taken (small, compressed, not regular FASTA, no memory):
// the caller parses on the host.  On success refs carries the names only (all that stream / filter print).
struct DeviceRefs { std::vector<char> names; std::vector<uint64_t> name_offsets; };
static std::map<std::string, rk_gzip*> g_gzip_refs; // -r files that are ordinary gzip (opened once; nullptr: looked at, not gzip)
static rk_gzip* gzip_ref_of(const char* path) {
    auto it = g_gzip_refs.find(path);
    if (it != g_gzip_refs.end()) return it->second;
    rk_gzip* gz = nullptr;
    if (rk_gzip_open(path, &gz) != RK_OK) gz = nullptr;
    g_gzip_refs[path] = gz;
    return gz;
}
static std::map<std::string, rk_bgzf*> g_bgzf_refs; // ... that are BGZF
static rk_bgzf* bgzf_ref_of(const char* path) {
    auto it = g_bgzf_refs.find(path);
    if (it != g_bgzf_refs.end()) return it->second;
    rk_bgzf* bz = nullptr;
    if (getenv("RKMH_BGZF") && atoi(getenv("RKMH_BGZF")) == 0) bz = nullptr;
    else if (rk_bgzf_open(path, &bz) != RK_OK) bz = nullptr;
    g_bgzf_refs[path] = bz;
    return bz;
}
// will refs_through_device take the -r files?  (sizes, total: the files' lengths and their sum with a newline after each)
static bool refs_for_device(const Opts& o, std::vector<int64_t>* sizes = nullptr, uint64_t* total_out = nullptr) {
    const char* env = getenv("RKMH_RAW_REFS");
    if (env && atoi(env) == 0) return false;
    const bool forced = env && atoi(env) == 1;
    std::vector<int64_t> size(o.refs.size(), 0);
    uint64_t total = 0;
    for (size_t i = 0; i < o.refs.size(); ++i) {
        if (!raw_eligible(o.refs[i], &size[i], '>')) {
            // an ordinary gzip file (genome.fa.gz as it is distributed): inflated on the device (rk_fasta_load_put_gzip); its text's
            // length is the trailer's word for it (a file of 4 GB of text or more ends up with the host parser)
            if (rk_bgzf* bz = bgzf_on_device() ? bgzf_ref_of(o.refs[i]) : nullptr) { // a bgzip'd genome: independent members (rk_fasta_load_put_bgzf)
                if (rk_bgzf_first_byte(bz) != '>') return false;
                size[i] = (int64_t)rk_bgzf_text_bytes(bz);
                total += (uint64_t)size[i] + 1;
                continue;
            }
            rk_gzip* gz = gzip_on_device() ? gzip_ref_of(o.refs[i]) : nullptr;
            if (!gz || rk_gzip_first_byte(gz) != '>') return false;
            size[i] = (int64_t)rk_gzip_text_bytes_hint(gz);
        }
        total += (uint64_t)size[i] + 1; // a '\n' after every file
    }
    if (o.refs.empty() || (!forced && total < ((uint64_t)64 << 20))) return false;
    if (sizes) *sizes = size;
    if (total_out) *total_out = total;
    return true;
}
static bool refs_through_device(RawEngine& eng, DeviceGroup& g, const Opts& o, int max_samples, uint64_t counter_slots, rk_seqset& refs,
                                DeviceRefs& keep) {
    std::vector<int64_t> size;
    uint64_t total = 0;
    if (!refs_for_device(o, &size, &total)) return false;
    eng.need_plain_workers = true;
    if (!eng.create(g)) return false;
    rk_fasta_load* load = nullptr;
    if (rk_fasta_load_create(g.ctx[0], total, &load) != RK_OK) {
        fprintf(stderr, "rkmh: references through the device: %s; parsing on the host\n", rk_last_error());
        return false;
    }
    struct Job { size_t file; int64_t lo, hi; uint64_t at; bool last; };
    std::vector<Job> jobs;
    struct GzRef { rk_gzip* gz; uint64_t at, size; };
    std::vector<GzRef> gz_refs;
    struct BzRef { rk_bgzf* bz; uint64_t at, size; };
    std::vector<BzRef> bz_refs;
    std::vector<int> fds(o.refs.size(), -1);
    {
        uint64_t at = 0;
        const int64_t B = (int64_t)eng.block;
        for (size_t i = 0; i < o.refs.size(); ++i) {
            if (rk_bgzf* bz = g_bgzf_refs.count(o.refs[i]) ? g_bgzf_refs[o.refs[i]] : nullptr) { // (refs_for_device found it to be BGZF)
                bz_refs.push_back(BzRef{bz, at, (uint64_t)size[i]});
                at += (uint64_t)size[i] + 1;
                continue;
            }
            if (rk_gzip* gz = g_gzip_refs.count(o.refs[i]) ? g_gzip_refs[o.refs[i]] : nullptr) { // (refs_for_device found it to be gzip)
                gz_refs.push_back(GzRef{gz, at, (uint64_t)size[i]});
                at += (uint64_t)size[i] + 1;
                continue;
            }
            fds[i] = open(o.refs[i], O_RDONLY);
            if (fds[i] < 0) { fprintf(stderr, "rkmh: cannot open %s\n", o.refs[i]); fail_exit(); }
            for (int64_t lo = 0; lo < size[i]; lo += B) {
                const int64_t hi = std::min(size[i], lo + B);
                jobs.push_back(Job{i, lo, hi, at + (uint64_t)lo, hi == size[i]});
            }
            at += (uint64_t)size[i] + 1;
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    auto work = [&](size_t wi) {
        if (eng.w[wi].dev != 0 || eng.w[wi].device_text) return; // the text goes to the device that sketches, through a page-locked text buffer
        if (!eng.w[wi].slot && rk_fastq_slot_create(g.ctx[0], eng.w[wi].bytes, &eng.w[wi].slot) != RK_OK) return;
        rk_fastq_slot* slot = eng.w[wi].slot;
        uint8_t* text = rk_fastq_slot_text(slot);
        for (size_t j = next.fetch_add(1); j < jobs.size() && !failed.load(); j = next.fetch_add(1)) {
            const Job& jb = jobs[j];
            int64_t have = 0;
            while (have < jb.hi - jb.lo) {
                const ssize_t n = pread(fds[jb.file], text + have, (size_t)(jb.hi - jb.lo - have), (off_t)(jb.lo + have));
                if (n <= 0) { fprintf(stderr, "rkmh: read error on %s\n", o.refs[jb.file]); fail_exit(); }
                have += n;
            }
            uint64_t nbytes = (uint64_t)have;
            if (jb.last) text[nbytes++] = '\n'; // (the slot holds 64 spare bytes)
            if (rk_fasta_load_put(load, slot, jb.at, nbytes) != RK_OK) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); failed = true; }
        }
    };
    double tr = now_s();
    std::vector<std::thread> th;
    for (size_t i = 0; i < eng.w.size(); ++i) th.emplace_back(work, i);
    for (const GzRef& gr : gz_refs) { // (this thread: a gzip stream is inflated stretch after stretch)
        uint64_t nb = 0;
        const int rc = failed.load() ? 1 : rk_fasta_load_put_gzip(load, gr.gz, gr.at, &nb);
        if (rc < 0) { fprintf(stderr, "rkmh: %s\n", rk_last_error()); fail_exit(); }
        if (rc != RK_OK || nb != gr.size || rk_fasta_load_put_newline(load, gr.at + nb) != RK_OK) failed = true; // (the host parser reads the references)
    }
    if (!bz_refs.empty() && !failed.load()) { // bgzip'd references: runs of members inflated in the buffers of one device-text slot made for the purpose
        const uint64_t job_text = (uint64_t)512 << 20;
        rk_fastq_slot* via = nullptr;
        if (rk_fastq_slot_create2(g.ctx[0], job_text + ((uint64_t)1 << 20), RK_SLOT_DEVICE_TEXT, &via) != RK_OK) failed = true;
        for (const BzRef& br : bz_refs) {
            if (failed.load()) break;
            std::vector<int64_t> first((size_t)rk_bgzf_members(br.bz) + 4);
            const int64_t nj = rk_bgzf_plan_members(br.bz, job_text - ((uint64_t)1 << 18), 16381, first.data(), (int64_t)first.size());
            if (nj < 0) { failed = true; break; }
            rk_host_register_readonly(rk_bgzf_image(br.bz), (size_t)rk_bgzf_file_bytes(br.bz)); // (the DMA engine reads the mapping itself; refused: staged uploads)
            for (int64_t j = 0; j < nj && !failed.load(); ++j) {
                const int rc = rk_fasta_load_put_bgzf(load, via, br.bz, first[(size_t)j], first[(size_t)j + 1], br.at + rk_bgzf_text_offset(br.bz, first[(size_t)j]));
                if (rc != RK_OK) failed = true; // (a damaged member as well: the host parser reports it)
            }
            if (!failed.load() && rk_fasta_load_put_newline(load, br.at + br.size) != RK_OK) failed = true;
        }
        if (via) rk_fastq_slot_destroy(via);
    }
    for (auto& t : th) t.join();
    for (int fd : fds) if (fd >= 0) close(fd);
    tick("references: text read and uploaded", tr);
    bool ok = !failed.load() && next.load() >= jobs.size();
    rk_fasta_index ix;
    memset(&ix, 0, sizeof ix);
    if (ok && rk_fasta_load_finish(load, total, &ix) != RK_OK) { fprintf(stderr, "rkmh: references through the device: %s; parsing on the host\n", rk_last_error()); ok = false; }
    if (ok && ix.status != 0) {
        if (g_timing) fprintf(stderr, "[rkmh timing] references: not plain line-structured FASTA (status %d): the host parser reads them\n", ix.status);
        ok = false;
    }
    tick("references: headers and line ends stripped on the device", tr);
    if (ok) {
        keep.name_offsets.assign(ix.name_offsets, ix.name_offsets + ix.nseq + 1);
        keep.names.assign(ix.names, ix.names + keep.name_offsets.back());
        keep.names.push_back('\0');
        CK(rk_set_references_fasta(g.ctx[0], load, o.ks.data(), (int)o.ks.size(), o.sketch, max_samples, counter_slots));
        memset(&refs, 0, sizeof refs);
        refs.nseq = ix.nseq;
        refs.names = keep.names.data();
        refs.name_offsets = keep.name_offsets.data();
        tick("references: sketched", tr);
        if (g_timing) fprintf(stderr, "[rkmh timing] references through the device: %lld sequences, %.0f MB of text\n", (long long)ix.nseq, (double)total / 1e6);
    }
    rk_fasta_load_destroy(load);
    return ok;
}

// the kseq-grammar scanner as a producer thread: batches of the given files (each from a byte offset, 0 = its start), numbered
static std::thread start_scanner(QueueT<Numbered>& q, std::vector<std::pair<const char*, uint64_t>> files) {
    return std::thread([&q, files] {
        int64_t seq = 0;
        for (auto& f : files) {
            rk_reader* rd = nullptr;
            if ((f.second ? rk_reader_open_at(f.first, f.second, &rd) : rk_reader_open(f.first, &rd)) != RK_OK) { q.err = rk_last_error(); break; }
            rk_reader_set_options(rd, RK_READER_NO_QUALS); // stream never looks at qualities
            for (;;) {
                Numbered nb;
                if (rk_reader_next(rd, 1 << 20, 1ull << 28, &nb.reads) != RK_OK) { q.err = rk_last_error(); break; }
                if (nb.reads.nseq == 0) { rk_seqset_free(&nb.reads); break; }
                nb.seq = seq++;
                q.push(nb);
            }
            rk_reader_close(rd);
            if (!q.err.empty()) break;
        }
        q.finish();
    });
}

static void run_scanner_pipeline(DeviceGroup& group, const rk_seqset& refs, const Opts& o, QueueT<Numbered>& q, std::thread& producer) {
    double t_cls = 0, t_emit = 0, t_wait = 0;
    // parser -> (one classify thread per device) -> writer.  Batches are numbered by the parser; the writer puts them back in
    // input order, so the output does not depend on how many devices took part or on which one was faster.
    if (q.cap < 2 * group.size()) { std::lock_guard<std::mutex> l(q.m); q.cap = 2 * group.size(); q.cv.notify_all(); }
    QueueT<Classified> done_q;
    done_q.cap = 2 * group.size() + 2;
    OutPool out_pool;
    std::thread writer([&] { // lines leave in read order: one writer, batches by number
        Classified c;
        std::string wbuf;
        std::map<int64_t, Classified> waiting;
        int64_t next = 0;
        while (done_q.pop(&c)) {
            waiting.emplace(c.seq, std::move(c));
            for (auto it = waiting.find(next); it != waiting.end(); it = waiting.find(next)) {
                double a = now_s();
                emit_lines(refs, it->second.reads, it->second.out4, o, wbuf);
                out_pool.put(it->second.out4, it->second.out_cap);
                rk_seqset_free(&it->second.reads);
                t_emit += now_s() - a;
                waiting.erase(it);
                ++next;
            }
        }
    });
    std::mutex tm;
    std::vector<std::string> werr(group.size());
    auto work = [&](size_t d) {
        for (;;) {
            double a = now_s();
            Numbered nb;
            if (!q.pop(&nb)) break;
            double b = now_s();
            Classified c;
            c.reads = nb.reads; c.seq = nb.seq;
            c.out4 = out_pool.get((size_t)c.reads.nseq, &c.out_cap);
            if (rk_classify_batch(group.ctx[d], c.reads.bases, c.reads.offsets, c.reads.nseq, c.out4) != RK_OK) { werr[d] = rk_last_error(); break; }
            double c2 = now_s();
            done_q.push(std::move(c));
            std::lock_guard<std::mutex> l(tm);
            t_wait += b - a; t_cls += c2 - b;
        }
    };
    std::vector<std::thread> workers;
    for (size_t d = 1; d < group.size(); ++d) workers.emplace_back(work, d);
    work(0);
    for (auto& t : workers) t.join();
    for (auto& e : werr) if (!e.empty()) { fprintf(stderr, "rkmh: %s\n", e.c_str()); fail_exit(); }
    done_q.finish();
    writer.join();
    if (g_timing) fprintf(stderr, "[rkmh timing] wait-for-parser %.3f s, classify %.3f s, format+write %.3f s (overlapped; summed over %zu device(s))\n", t_wait, t_cls, t_emit, group.size());
    producer.join();
    if (!q.err.empty()) { fprintf(stderr, "rkmh: %s\n", q.err.c_str()); fail_exit(); }
}

static int main_stream(int argc, char** argv) {
    Opts o;
    const char* pre_refs = nullptr;
    const char* read_map = nullptr;
    if (argc <= 2) { help_stream(); exit(1); }
    static struct option long_options[] = {
        {"help", no_argument, 0, 'h'},           {"kmer", required_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'},    {"reference", required_argument, 0, 'r'},
        {"sketch-size", required_argument, 0, 's'}, {"ref-sketch", required_argument, 0, 'S'},
        {"threads", required_argument, 0, 't'},  {"min-kmer-occurence", required_argument, 0, 'M'},
        {"min-matches", required_argument, 0, 'N'}, {"min-diff", required_argument, 0, 'D'},
        {"max-samples", required_argument, 0, 'I'}, {"pre-reads", required_argument, 0, 'F'},
        {"pre-references", required_argument, 0, 'R'}, {"read-kmer-map-file", required_argument, 0, 'p'},
        {"ref-kmer-map-file", required_argument, 0, 'q'}, {"in-stream", no_argument, 0, 'i'},
        {"output-reads", no_argument, 0, 'z'},   {"merge-sketch", no_argument, 0, 'm'},
        {"device", required_argument, 0, 1000},  {"depth-map-cache", required_argument, 0, 1001}, {"kmer-cache", required_argument, 0, 1003},
        {"devices", required_argument, 0, 1002}, {"no-kmer-cache", no_argument, 0, 1005}, HASH_POLICY_OPTION, {0, 0, 0, 0}};
    optind = 2;
    int c;
    while ((c = getopt_long(argc, argv, "zmhdk:f:r:s:S:t:M:N:I:R:F:p:q:iD:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 'm': case 'i': case 'z': break;                 // parsed and ignored, rkmh.cpp:656-658,709-714
            case 'R': pre_refs = optarg; break;               // pre-hashed references: parsed but unimplemented in the reference (:662-664)
            // -p/-q (k-mer map files): the reference parses them and does nothing (bodies commented out, :665-670, :744-769);
            // so do we -- no file is read or written.  The reusable depth map is this build's own, explicit option below.
            case 'F': o.packed.push_back(optarg); break;      // --pre-reads: parsed and unused in the reference (:659-664); here: reads packed by `rkmh pack`
            case 'p': case 'q': case 'S': break; // parsed, bodies empty in the reference (:665-670,:697-700)
            case 1001: read_map = optarg; break;              // --depth-map-cache FILE (not a reference flag): see the -M block
            case 1003: o.kmer_cache = optarg; break;          // --kmer-cache FILE (not a reference flag): rk_set_kmer_cache
            case 1005: g_no_kmer_cache = true; break;
            case 't': o.threads = atoi(optarg); break;
            case 'r': o.refs.push_back(optarg); break;
            case 'f': o.reads.push_back(optarg); break;
            case 'k': o.ks.push_back(atoi(optarg)); break;
            case 'N': o.min_matches = atoi(optarg); break;
            case 'D': o.min_diff = atoi(optarg); break;
            case 's': o.sketch = atoi(optarg); break;
            case 'M': o.min_occ = atoi(optarg); o.read_depth = true; break;
            case 'I': o.max_samples = atoi(optarg); o.ref_depth = true; break;
            case 1000: o.device = atoi(optarg); break;
            case 1002: o.devices = parse_devices(optarg); break;
            case '?': case 'h': default: print_help(); exit(1);
        }
    }
    LoadedSketches pre;
    if (pre_refs) {
        if (!load_sketch_json(pre_refs, pre)) { fprintf(stderr, "rkmh: cannot load sketches from %s\n", pre_refs); exit(1); }
        // sketches hashed under another policy would meet read hashes they can never equal: refused, not classified against
        rk_policy theirs;
        rk_default_policy(&theirs);
        if (rk_policy_parse(pre.policy.c_str(), &theirs) != RK_OK) { fprintf(stderr, "rkmh: %s: %s\n", pre_refs, rk_last_error()); exit(1); }
        if (!rk_policy_same_hashes(&theirs, &g_policy)) {
            fprintf(stderr, "rkmh: %s holds sketches hashed with %s, this run hashes with %s: pass --hash-policy %s\n", pre_refs,
                    policy_text(theirs).c_str(), policy_text(g_policy).c_str(), policy_text(theirs).c_str());
            exit(1);
        }
        o.ks = pre.ks; o.sketch = pre.S;
    }
    if (o.ks.empty()) {
        fprintf(stderr, "No kmer size(s) provided. Will use a default kmer size of 16.\n"); // rkmh.cpp:729
        o.ks.push_back(16);
    }
    if (o.refs.empty() && !pre_refs) { fprintf(stderr, "rkmh: at least one -r reference file (or -R sketches) is required\n"); exit(1); }
    if (!o.packed.empty() && !o.reads.empty()) { fprintf(stderr, "rkmh: give the reads either as text (-f) or as packed files (-F), not both\n"); exit(1); }

    // Which front end reads the reads: regular uncompressed FASTQ files go through the device (stream_file_raw); everything else --
    // gzip, STDIN, FASTA, -M (which needs all reads twice) -- through the kseq-grammar scanner.  RKMH_RAW=0 forces the scanner.
    g_read_paths = &o.reads;
    std::vector<int64_t> raw_size(o.reads.size(), -1);
    bool any_raw = false, all_raw = !o.reads.empty();
    if (!(getenv("RKMH_RAW") && atoi(getenv("RKMH_RAW")) == 0))
        for (size_t i = 0; i < o.reads.size(); ++i) { if (raw_eligible(o.reads[i], &raw_size[i])) any_raw = true; else { raw_size[i] = -1; all_raw = false; } }
    else all_raw = false;
    if (o.read_depth) any_raw = false; // -M reads every file twice: device front end only when ALL files qualify (two_pass_raw)
    // The scanner starts NOW when it has all the files (streaming path): while the GPU contexts come up and the references are
    // sketched -- a few tenths of a second -- it is already filling its first batches.
    QueueT<Numbered> q;
    q.cap = 4;
    std::thread producer;
    if (!o.read_depth && !any_raw && o.packed.empty()) {
        std::vector<std::pair<const char*, uint64_t>> files;
        for (const char* path : o.reads) files.emplace_back(path, 0);
        producer = start_scanner(q, files);
    }

    double t0 = now_s();
    DeviceGroup group;
    group.create(o);
    if (o.read_depth) for (rk_ctx* cx : group.ctx) CK(rk_set_min_num_bound(cx, min_num_bound_for(o.min_matches)));
    rk_ctx* ctx = group.ctx[0];
    tick("context", t0);
    // the front end's kernels (and the inflater's) are loaded while the references are sketched, not in front of the first block
    // ... and so are the front end's engine and its first slot made and the BGZF mappings page-locked (unless the references themselves
    // go through the engine: then it is made for them first)
    RawEngine eng;       // the workers and page-locked buffers of the device front ends (created by whoever needs them first)
    std::thread warm;
    const bool raw_run = any_raw || (o.read_depth && all_raw && !read_map);
    if (raw_run && !(getenv("RKMH_WARM_UP") && atoi(getenv("RKMH_WARM_UP")) == 0)) {
        const bool prepare = !pre_refs ? !refs_for_device(o) : true;
        warm = std::thread([&o, &eng, &group, prepare] {
            const std::vector<int> ids = o.devices.empty() ? std::vector<int>{o.device} : o.devices;
            for (int id : ids) rk_warm_up(id, (!g_bgzf.empty() || !g_gzip.empty()) && bgzf_on_device());
            if (prepare && eng.create(group)) register_bgzf_mappings();
        });
    }
    rk_seqset refs;
    memset(&refs, 0, sizeof refs);
    DeviceRefs dev_refs;
    bool refs_owned = !pre_refs;
    std::string pre_names;
    std::vector<uint64_t> pre_noff;
    if (pre_refs) { // names come from the JSON file; emit_lines only needs names + name_offsets
        pre_noff.push_back(0);
        for (auto& nm : pre.names) { pre_names += nm; pre_names += '\0'; pre_noff.push_back(pre_names.size()); }
        refs.nseq = (int64_t)pre.names.size();
        refs.names = &pre_names[0];
        refs.name_offsets = pre_noff.data();
        CK(rk_set_reference_sketches(ctx, pre.sk.data(), pre.lens.data(), (int)pre.lens.size(), o.ks.data(), (int)o.ks.size(), o.sketch));
    } else if (refs_through_device(eng, group, o, o.ref_depth ? o.max_samples : -1, 0, refs, dev_refs)) {
        refs_owned = false;
    } else {
        CK(rk_parse_files(o.refs.data(), (int)o.refs.size(), &refs));
        if (refs.nseq < 1) { fprintf(stderr, "rkmh: no reference sequences found\n"); exit(1); }
        CK(rk_set_references(ctx, refs.bases, refs.offsets, (int)refs.nseq, o.ks.data(), (int)o.ks.size(), o.sketch,
                             o.ref_depth ? o.max_samples : -1, 0));
    }
    group.share_references(o);
    tick("references", t0);
    if (warm.joinable()) { warm.join(); tick("kernels loaded (waited)", t0); }
    if (!o.packed.empty()) { // reads written by `rkmh pack`: nothing to parse
        run_packed(group, refs, o, o.packed, RAW_STREAM, 200000000ull, min_num_bound_for(o.min_matches), t0);
        fflush(stdout);
        tick("main loop + flush", t0);
        done_exit();
    }
    std::string buf;
    std::vector<int32_t> out4;
    bool depth_done = false;
    std::vector<rk_counter*> cnts;
    const bool compact_ok = o.read_depth && compact_maps_wanted(min_num_bound_for(o.min_matches), read_map);
    if (o.read_depth) {
        make_depth_maps(group, 200000000ull, compact_ok && all_raw, cnts); // HASHTCounter(200000000), rkmh.cpp:739
        tick("depth tables", t0);
    }
    if (o.read_depth && all_raw && !read_map) {
        // regular FASTQ files: both passes through the device front end, the reads are never held in host memory
        if (eng.create(group)) {
            depth_done = two_pass_raw(eng, group, refs, o, raw_size, cnts, RAW_STREAM, t0, 200000000ull);
            if (g_timing) fprintf(stderr, "[rkmh timing] device front end: %lld blocks, %lld records; read %.3f s, device %.3f s, format %.3f s (summed over %zu workers, both passes)\n",
                                  (long long)eng.blocks, (long long)eng.records, eng.t_read, eng.t_dev, eng.t_fmt, eng.w.size());
        }
    }
    if (o.read_depth && depth_done) {
        for (rk_counter* k : cnts) rk_counter_destroy(k);
    } else if (o.read_depth) {
        // two passes over ALL reads (rkmh.cpp:904-948): the reference holds them in RAM, so do we
        rk_seqset reads;
        CK(rk_parse_files(o.reads.data(), (int)o.reads.size(), &reads));
        const bool cmp = compact_ok && reads_fit_sketch(reads, o);
        if (cmp != (rk_counter_is_compact(cnts[0]) != 0)) make_depth_maps(group, 200000000ull, cmp, cnts);
        rk_counter* cnt = cnts[0];
        // --depth-map-cache FILE: reuse a saved depth map (pass 1 is skipped) or save this run's for the next one.  The file
        // records what it was counted from (k list, hashing policy, fingerprint of the read set); a file that does not match
        // THIS run is refused with a diagnostic rather than used (CK exits).
        uint8_t tag[RK_DEPTH_TAG_BYTES];
        if (read_map) CK(rk_depth_map_tag(ctx, o.ks.data(), (int)o.ks.size(), reads.bases, reads.offsets, reads.nseq, tag));
        FILE* probe = read_map ? fopen(read_map, "rb") : nullptr;
        out4.resize((size_t)reads.nseq * 4);
        if (probe) {
            fclose(probe);
            CK(rk_counter_load_tagged(cnt, read_map, tag, sizeof tag));
            two_pass_on_group(group, reads, 200000000ull, o.min_occ, cnts, false, out4.data());
        } else if (read_map) { // pass 1 alone first: the summed table is saved before the masked pass
            std::vector<std::string> err(group.size());
            std::vector<std::thread> th;
            for (size_t d = 0; d < group.size(); ++d)
                th.emplace_back([&, d] {
                    const int64_t lo = share_lo(reads.nseq, d, group.size()), hi = share_lo(reads.nseq, d + 1, group.size());
                    if (rk_count_batch(group.ctx[d], reads.bases, reads.offsets + lo, hi - lo, cnts[d]) != RK_OK) err[d] = rk_last_error();
                });
            for (auto& t : th) t.join();
            for (auto& e : err) if (!e.empty()) { fprintf(stderr, "rkmh: %s\n", e.c_str()); exit(1); }
            for (size_t d = 1; d < group.size(); ++d) CK(rk_counter_add(cnt, cnts[d]));
            CK(rk_counter_save_tagged(cnt, read_map, tag, sizeof tag));
            two_pass_on_group(group, reads, 200000000ull, o.min_occ, cnts, false, out4.data());
        } else two_pass_on_group(group, reads, 200000000ull, o.min_occ, cnts, true, out4.data());
        emit_lines(refs, reads, out4.data(), o, buf);
        for (rk_counter* k : cnts) rk_counter_destroy(k);
        rk_seqset_free(&reads);
    } else {
        if (!any_raw) run_scanner_pipeline(group, refs, o, q, producer);
        else {
            const bool eng_ok = eng.create(group);
            tick("device front end", t0);
            for (size_t i = 0; i < o.reads.size();) {
                int64_t resume = 0;
                if (eng_ok && raw_size[i] >= 0) { // the run of files from here on that the device front end reads, as one pipeline
                    size_t j = i;
                    while (j < o.reads.size() && raw_size[j] >= 0) ++j;
                    const std::vector<const char*> run(o.reads.begin() + (long)i, o.reads.begin() + (long)j);
                    const std::vector<int64_t> sizes(raw_size.begin() + (long)i, raw_size.begin() + (long)j);
                    size_t ff = 0;
                    resume = stream_files_raw(eng, group, refs, o, run, sizes, RAW_STREAM, nullptr, &ff);
                    if (resume < 0) { i = j; continue; }
                    i += ff; // (the files in front of the refused block are done)
                    fflush(stdout);
                    if (g_timing) fprintf(stderr, "[rkmh timing] %s: not four lines per record at byte %lld: the scanner reads on from there\n", o.reads[i], (long long)resume);
                }
                QueueT<Numbered> q1;
                q1.cap = 4;
                std::thread p1 = start_scanner(q1, {{o.reads[i], (uint64_t)resume}});
                run_scanner_pipeline(group, refs, o, q1, p1);
                ++i;
            }
            if (g_timing) fprintf(stderr, "[rkmh timing] device front end: %lld blocks, %lld records; read %.3f s, upload + index + classify %.3f s, format %.3f s (summed over %zu workers)\n",
                                  (long long)eng.blocks, (long long)eng.records, eng.t_read, eng.t_dev, eng.t_fmt, eng.w.size());
        }
    }
    fflush(stdout);
    tick("main loop + flush", t0);
    // everything is written and the process ends here: releasing page-locked buffers, streams and contexts one by one took 0.08 s of
    // a 0.55 s run and produces nothing (the operating system takes it all back at once); a profiler's run keeps the orderly way out
    if (getenv("RKMH_SLOW_EXIT")) {
        eng.destroy();
        if (refs_owned) rk_seqset_free(&refs);
        group.destroy();
        tick("teardown", t0);
    }
    done_exit();
}

// filter: main_filter, src/rkmh.cpp:996-1424.  Same sketches as stream; the decision is filter_decide (above).
static void help_filter() {
    fprintf(stderr,
            "rkmh filter -r <refs.fa> -f <reads.fq> [-k <k>]... [-s <sketch>] [-M n] [-I n] [-N n] [-D n] [-i]\n"
            "  prints the reads (as >name / SEQ / + / QUAL) whose best reference passes the match and diff filters;\n"
            "  -i then classifies reads arriving on STDIN and prints one 'Sample: ... Result: ...' line each\n" HASH_POLICY_HELP);
}

static int main_filter(int argc, char** argv) {
    Opts o;
    bool in_stream = false;
    if (argc <= 2) { help_filter(); exit(1); }
    static struct option long_options[] = {
        {"help", no_argument, 0, 'h'},           {"kmer", required_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'},    {"reference", required_argument, 0, 'r'},
        {"sketch-size", required_argument, 0, 's'}, {"ref-sketch", required_argument, 0, 'S'},
        {"threads", required_argument, 0, 't'},  {"min-kmer-occurence", required_argument, 0, 'M'},
        {"min-matches", required_argument, 0, 'N'}, {"min-diff", required_argument, 0, 'D'},
        {"max-samples", required_argument, 0, 'I'}, {"pre-reads", required_argument, 0, 'F'},
        {"pre-references", required_argument, 0, 'R'}, {"read-kmer-map-file", required_argument, 0, 'p'},
        {"ref-kmer-map-file", required_argument, 0, 'q'}, {"in-stream", no_argument, 0, 'i'},
        {"device", required_argument, 0, 1000}, {"devices", required_argument, 0, 1002}, {"kmer-cache", required_argument, 0, 1003}, {"no-kmer-cache", no_argument, 0, 1005}, HASH_POLICY_OPTION, {0, 0, 0, 0}};
    optind = 2;
    int c;
    while ((c = getopt_long(argc, argv, "hdk:f:r:s:S:t:M:N:I:R:F:p:q:iD:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1005: g_no_kmer_cache = true; break;
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 1003: o.kmer_cache = optarg; break;
            case 'F': o.packed.push_back(optarg); break;      // reads packed by `rkmh pack` (--pre-reads is parsed and unused in the reference, rkmh.cpp:1139-1141)
            case 'R': case 'p': case 'q': case 'S': break; // parsed, bodies empty (rkmh.cpp:1142-1151)
            case 't': o.threads = atoi(optarg); break;
            case 'r': o.refs.push_back(optarg); break;
            case 'f': o.reads.push_back(optarg); break;
            case 'k': o.ks.push_back(atoi(optarg)); break;
            case 'N': o.min_matches = atoi(optarg); break;
            case 'D': o.min_diff = atoi(optarg); break;
            case 's': o.sketch = atoi(optarg); break;
            case 'M': o.min_occ = atoi(optarg); o.read_depth = true; break;
            case 'I': o.max_samples = atoi(optarg); o.ref_depth = true; break;
            case 'i': in_stream = true; break;
            case 1000: o.device = atoi(optarg); break;
            case 1002: o.devices = parse_devices(optarg); break;
            case '?': case 'h': default: print_help(); exit(1);
        }
    }
    if (o.ks.empty()) {
        fprintf(stderr, "No kmer size(s) provided. Will use a default kmer size of 16.\n");
        o.ks.push_back(16);
    }
    if (o.refs.empty()) { fprintf(stderr, "rkmh: at least one -r reference file is required\n"); exit(1); }
    if (!o.packed.empty() && (!o.reads.empty() || in_stream)) { fprintf(stderr, "rkmh: give the reads either as text (-f / -i) or as packed files (-F), not both\n"); exit(1); }
    // Regular uncompressed FASTQ files and BGZF files go through the device front end (stream_file_raw / two_pass_raw): the host neither
    // parses the reads nor holds them.  RKMH_RAW=0, plain gzip, FASTA and text that is not four lines per record take the parse-everything
    // path.  (Looked at before the engine is made: it is laid out for the kinds of read files there are.)
    g_read_paths = &o.reads;
    std::vector<int64_t> raw_size(o.reads.size(), -1);
    bool all_raw = !o.reads.empty() && !(getenv("RKMH_RAW") && atoi(getenv("RKMH_RAW")) == 0);
    for (size_t i = 0; all_raw && i < o.reads.size(); ++i) all_raw = raw_eligible(o.reads[i], &raw_size[i]);
    double t0 = now_s();
    DeviceGroup group;
    group.create(o);
    RawEngine eng;       // the workers and page-locked buffers of the device front ends
    // (it is made, with its first slot, and the BGZF mappings are page-locked while the references are read and sketched -- unless those go through the engine themselves)
    std::thread prep;
    if (all_raw && !refs_for_device(o) && !(getenv("RKMH_WARM_UP") && atoi(getenv("RKMH_WARM_UP")) == 0))
        prep = std::thread([&o, &eng, &group] {
            const std::vector<int> ids = o.devices.empty() ? std::vector<int>{o.device} : o.devices;
            for (int id : ids) rk_warm_up(id, (!g_bgzf.empty() || !g_gzip.empty()) && bgzf_on_device());
            if (eng.create(group)) register_bgzf_mappings();
        });
    // file mode compares read_min_lens with 0 (rkmh.cpp:1292); the STDIN lines print min(len) itself (:1397): exact there
    // (with -D >= 0 a read that shares nothing fails the diff test anyway, so not even min(read_min_lens, 1) is needed: bound 0)
    const int filter_bound = (o.read_depth && !in_stream) ? min_num_bound_for(o.min_diff >= 0 ? -1 : 0) : -1;
    if (o.read_depth && !in_stream) for (rk_ctx* cx : group.ctx) CK(rk_set_min_num_bound(cx, filter_bound));
    rk_ctx* ctx = group.ctx[0];
    tick("context", t0);
    rk_seqset refs;
    memset(&refs, 0, sizeof refs);
    DeviceRefs dev_refs;
    // reference sketches: the sample-count filter applies when max_samples < 100000 (rkmh.cpp:1211); its counter is
    // filled once per distinct hash per reference and only when -I was given (rkmh.cpp:1193, :348-355); 10 M slots (:1188)
    CK(rk_set_reference_count_mode(ctx, 1));
    const bool ref_filter = o.max_samples < 100000;
    if (refs_through_device(eng, group, o, ref_filter ? o.max_samples : -1, 10000000ull, refs, dev_refs)) tick("references: upload + strip + sketch on the device", t0);
    else {
        CK(rk_parse_files(o.refs.data(), (int)o.refs.size(), &refs));
        if (refs.nseq < 1) { fprintf(stderr, "rkmh: no reference sequences found\n"); exit(1); }
        tick("parse references", t0);
        CK(rk_set_references(ctx, refs.bases, refs.offsets, (int)refs.nseq, o.ks.data(), (int)o.ks.size(), o.sketch,
                             ref_filter ? o.max_samples : -1, 10000000ull));
    }
    std::vector<int32_t> ref_lens((size_t)refs.nseq);
    {
        std::vector<uint64_t> sk((size_t)refs.nseq * (size_t)o.sketch);
        CK(rk_get_reference_sketches(ctx, sk.data(), ref_lens.data()));
    }
    group.share_references(o);
    tick("sketch references", t0);
    if (prep.joinable()) prep.join();
    if (!o.packed.empty()) { // reads written by `rkmh pack`: nothing to parse
        run_packed(group, refs, o, o.packed, RAW_FILTER, 10000000ull, filter_bound, t0);
        fflush(stdout);
        done_exit();
    }
    std::vector<rk_counter*> cnts;
    const bool compact_ok = o.read_depth && !in_stream && compact_maps_wanted(filter_bound, nullptr);
    std::string buf;
    std::vector<int32_t> out4;
    // filter's records of a parsed batch (rkmh.cpp:1292-1300)
    auto emit_passing = [&](const rk_seqset& reads, const int32_t* rows) {
        for (int64_t i = 0; i < reads.nseq; ++i) {
            const int32_t* r = rows + (size_t)i * 4;
            const FilterDecision d = filter_decide(r, o.min_diff);
            // rkmh.cpp:1292-1293.  read_min_lens <= 0 implies shared == 0 (a shared hash is a min), so the conjunction is the same
            // predicate on exact rows -- and it stays right on rows whose min_num was clamped to 0 (bound 0, only with -D >= 0)
            const bool depth_filter = r[3] <= 0 && d.shared <= 0, match_filter = d.shared < o.min_matches;
            if (depth_filter || match_filter || !d.diff_ok) continue;
            buf += '>';
            buf += reads.names + reads.name_offsets[i];
            buf += '\n';
            for (uint64_t j = reads.offsets[i]; j < reads.offsets[i + 1]; ++j) {
                signed char ch = (signed char)reads.bases[j];
                buf += (char)(((int)ch - 91) > 0 ? ch - 32 : ch); // to_upper as parse_fastas applies it (rkmh.cpp:280)
            }
            buf += "\n+\n";
            if (reads.quals) buf.append(reads.quals + reads.offsets[i], (size_t)(reads.offsets[i + 1] - reads.offsets[i]));
            buf += '\n';
            if (buf.size() > (1u << 22)) { fwrite(buf.data(), 1, buf.size(), stdout); buf.clear(); }
        }
        fwrite(buf.data(), 1, buf.size(), stdout);
        buf.clear();
    };
    bool files_done = o.reads.empty();
    if (all_raw) {
        if (eng.create(group)) {
            if (o.read_depth) {
                make_depth_maps(group, 10000000ull, compact_ok, cnts); // read_hash_counter, rkmh.cpp:1187
                files_done = two_pass_raw(eng, group, refs, o, raw_size, cnts, RAW_FILTER, t0, 10000000ull);
            }
            else {
                for (size_t i = 0; i < o.reads.size(); ++i) {
                    const std::vector<const char*> run(o.reads.begin() + (long)i, o.reads.end());
                    const std::vector<int64_t> sizes(raw_size.begin() + (long)i, raw_size.end());
                    size_t ff = 0;
                    const int64_t resume = stream_files_raw(eng, group, refs, o, run, sizes, RAW_FILTER, nullptr, &ff);
                    if (resume < 0) break;
                    i += ff; // (the files in front of the refused block are done; this one goes on through the scanner)
                    fflush(stdout);
                    // the rest of this file through the kseq-grammar scanner, batch by batch
                    if (g_timing) fprintf(stderr, "[rkmh timing] %s: not four lines per record at byte %lld: the scanner reads on from there\n", o.reads[i], (long long)resume);
                    rk_reader* rd = nullptr;
                    CK(rk_reader_open_at(o.reads[i], (uint64_t)resume, &rd));
                    for (;;) {
                        rk_seqset part;
                        CK(rk_reader_next(rd, 1 << 20, 1ull << 28, &part));
                        if (part.nseq == 0) { rk_seqset_free(&part); break; }
                        out4.resize((size_t)part.nseq * 4);
                        CK(rk_classify_batch(ctx, part.bases, part.offsets, part.nseq, out4.data()));
                        emit_passing(part, out4.data());
                        rk_seqset_free(&part);
                    }
                    rk_reader_close(rd);
                }
                fflush(stdout);
                files_done = true;
                tick("device front end + classify + emit", t0);
            }
            if (g_timing) fprintf(stderr, "[rkmh timing] device front end: %lld blocks, %lld records; read %.3f s, device %.3f s, format %.3f s (summed over %zu workers)\n",
                                  (long long)eng.blocks, (long long)eng.records, eng.t_read, eng.t_dev, eng.t_fmt, eng.w.size());
        }
    }
    if (!files_done) {
        rk_seqset reads;
        CK(rk_parse_files(o.reads.data(), (int)o.reads.size(), &reads));
        tick("parse reads", t0);
        out4.resize((size_t)reads.nseq * 4);
        if (o.read_depth) {
            // count (rkmh.cpp:321-338), then keep get(h) >= min_kmer_occ (:1260); the reads are spread over the devices, the
            // depth tables summed in between
            const bool cmp = compact_ok && reads_fit_sketch(reads, o);
            if (cnts.empty() || cmp != (rk_counter_is_compact(cnts[0]) != 0)) make_depth_maps(group, 10000000ull, cmp, cnts);
            else group_run(group, [&](size_t d) { return rk_counter_clear(cnts[d]); });
            two_pass_on_group(group, reads, 10000000ull, o.min_occ, cnts, true, out4.data());
        } else {
            group_run(group, [&](size_t d) {
                const int64_t lo = share_lo(reads.nseq, d, group.size()), hi = share_lo(reads.nseq, d + 1, group.size());
                return rk_classify_batch(group.ctx[d], reads.bases, reads.offsets + lo, hi - lo, out4.data() + lo * 4);
            });
        }
        tick("count + classify", t0);
        emit_passing(reads, out4.data());
        tick("emit", t0);
        rk_seqset_free(&reads);
    }
    if (in_stream) { // rkmh.cpp:1329-1408: reads from STDIN are classified, one line each
        // (-i keeps exact rows and full tables -- its lines print min(len) itself -- so the table, if any, is the one the files filled)
        if (cnts.empty()) make_depth_maps(group, 10000000ull, false, cnts);
        rk_counter* cnt = cnts[0];
        CK(rk_set_depth_filter(ctx, o.min_occ > 0 ? cnt : nullptr, o.min_occ)); // :1365
        rk_reader* rd = nullptr;
        CK(rk_reader_open("-", &rd));
        char line[8192];
        for (;;) {
            rk_seqset s;
            CK(rk_reader_next(rd, 1 << 18, 1ull << 27, &s));
            if (s.nseq == 0) { rk_seqset_free(&s); break; }
            out4.resize((size_t)s.nseq * 4);
            CK(rk_classify_batch(ctx, s.bases, s.offsets, s.nseq, out4.data()));
            for (int64_t i = 0; i < s.nseq; ++i) {
                const int32_t* r = &out4[(size_t)i * 4];
                const FilterDecision d = filter_decide(r, o.min_diff);
                const int uni = d.ref < 0 ? 0 : (r[3] < ref_lens[(size_t)d.ref] ? r[3] : ref_lens[(size_t)d.ref]);
                int n = snprintf(line, sizeof line, "Sample: %s\tResult: %s\t%d\t%d\t%s\t%s\t%s\n", s.names + s.name_offsets[i],
                                 d.ref < 0 ? "" : refs.names + refs.name_offsets[d.ref], d.shared, uni, r[3] <= 0 ? "FAIL:DEPTH" : "",
                                 d.shared < o.min_matches ? "FAIL:MATCHES" : "", d.diff_ok ? "" : "FAIL:DIFF");
                if (n > 0) buf.append(line, (size_t)(n < (int)sizeof line ? n : (int)sizeof line - 1));
            }
            fwrite(buf.data(), 1, buf.size(), stdout);
            buf.clear();
            rk_seqset_free(&s);
        }
        rk_reader_close(rd);
    }
    // everything is written: the process ends here, as in main_stream -- freeing a genome-sized reference set, the contexts and the
    // HIP runtime's exit handlers took 0.7 s of a 2.5 s C4 run and produce nothing
    done_exit();
}

// call: main_call, src/rkmh.cpp:1455-1904.  The GPU returns one record per candidate k-mer that passed the depth
// tests; the VCF rows are the records aggregated by (ref, pos, orig, alt) exactly as rkmh.cpp:1821-1829 / :1856-1863.
static void help_call() {
    fprintf(stderr,
            "rkmh call -r <ref.fa> -f <reads.fq> [-k <k>] [-w <window>]\n"
            "  calls SNPs and 1-bp deletions from the k-mer depth of the reads along the reference (VCF-like rows)\n" HASH_POLICY_HELP);
}
#include <map>
static int main_call(int argc, char** argv) {
    std::vector<const char*> refs, reads;
    std::vector<int> ks;
    int window_len = 100, device = 0;
    bool show_depth = false;
    if (argc <= 2) { help_call(); exit(1); }
    static struct option long_options[] = {
        {"help", no_argument, 0, 'h'},        {"kmer", required_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'}, {"reference", required_argument, 0, 'r'},
        {"sketch", required_argument, 0, 's'}, {"threads", required_argument, 0, 't'},
        {"window-len", required_argument, 0, 'w'}, {"show-depth", no_argument, 0, 'd'},
        {"device", required_argument, 0, 1000}, HASH_POLICY_OPTION, {0, 0, 0, 0}};
    optind = 2;
    int c;
    while ((c = getopt_long(argc, argv, "hdk:f:r:s:t:w:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 'r': refs.push_back(optarg); break;
            case 'f': reads.push_back(optarg); break;
            case 'k': ks.push_back(atoi(optarg)); break;
            case 'w': window_len = atoi(optarg); break;
            case 'd': show_depth = true; break;             // sets show_depth, clears output_vcf: prints nothing (Appendix A.3)
            case 's': case 't': break;                        // parsed, unused
            case 1000: device = atoi(optarg); break;
            case '?': case 'h': default: print_help(); exit(1);
        }
    }
    if (ks.empty()) {
        fprintf(stderr, "No kmer size(s) provided. Will use a default kmer size of 16.\n");
        ks.push_back(16);
    } else if (ks.size() > 1) {                               // rkmh.cpp:1543-1552
        fprintf(stderr, "Only a single kmer size may be used for calling.\nSizes provided: ");
        for (int k : ks) fprintf(stderr, "%d ", k);
        fprintf(stderr, "\nPlease choose a single kmer size.\n");
        exit(1);
    }
    fprintf(stderr, "Parsing sequences...\n");
    if (refs.empty()) {
        fprintf(stderr, "No references were provided. Please provide at least one reference file in fasta/fastq format.\n");
        help_call(); exit(1);
    }
    if (reads.empty()) {
        fprintf(stderr, "No reads were provided. Please provide at least one read file in fasta/fastq format.\n");
        help_call(); exit(1);
    }
    rk_seqset R, Q;
    CK(rk_parse_files(refs.data(), (int)refs.size(), &R));
    CK(rk_parse_files(reads.data(), (int)reads.size(), &Q));
    if (R.nseq < 1) { fprintf(stderr, "rkmh: no reference sequences found\n"); exit(1); }
    if (R.nseq > 1) fprintf(stderr, "WARNING: more than one ref provided. VCF will not be correct\n");
    if (show_depth) return 0;
    rk_ctx* ctx = nullptr;
    CK(rk_ctx_create(device, &g_policy, &ctx));
    rk_call_record* rec = nullptr;
    int64_t nrec = 0;
    CK(rk_call(ctx, R.bases, R.offsets, (int)R.nseq, Q.bases, Q.offsets, Q.nseq, ks[0], window_len, &rec, &nrec));
    printf("##fileformat=VCF4.2\n##source=rkmh\n##reference=%s\n"
           "##INFO=<ID=KD,Number=1,Type=Integer,Description=\"Number of times call for specific kmer appears\">\n"
           "##INFO=<ID=MD,Number=1,Type=Integer,Description=\"Maximum depth found for the rescue kmer.\">\n"
           "##INFO=<ID=RD,Number=1,Type=Integer,Description=\"Average depth in region\">"
           "##INFO=<ID=OD,Number=1,Type=Integer,Description=\"Depth of original kmer at site before modification.\">\n", refs[0]);
    struct Agg { int kc = 0, md = 0, rd = 0, od = 0; };
    std::map<std::string, Agg> rows; // lexicographic key order, as the reference's std::map (rkmh.cpp:1885)
    char key[4096];
    for (int64_t i = 0; i < nrec; ++i) {
        const rk_call_record& r = rec[i];
        snprintf(key, sizeof key, "%s\t%d\t.\t%c\t%c", R.names + R.name_offsets[r.ref], r.pos, (char)r.orig, (char)r.alt);
        Agg& a = rows[key];
        a.kc += 1;
        if (r.alt_depth > a.md) a.md = r.alt_depth;
        if (r.avg_d > a.rd) a.rd = r.avg_d;
        if (r.depth > a.od) a.od = r.depth;
    }
    for (auto& kv : rows) printf("%s\t99\tPASS\tKC=%d;MD=%d;RD=%d;OD=%d\n", kv.first.c_str(), kv.second.kc, kv.second.md, kv.second.rd, kv.second.od);
    fflush(stdout);
    rk_free(rec);
    rk_seqset_free(&R); rk_seqset_free(&Q);
    rk_ctx_destroy(ctx);
    return 0;
}

// ---- JSON sketches (the schema of dump_hash_json, src/rkmh.cpp:489-525; dead code in the reference, kept here as the
// interchange format SURVEY.md section 8f ranks next).  Keys are emitted in the alphabetical order nlohmann::json uses.
static void json_escape(std::string& out, const char* s) {
    for (; *s; ++s) {
        unsigned char ch = (unsigned char)*s;
        if (ch == '"' || ch == '\\') { out += '\\'; out += (char)ch; }
        else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); out += b; }
        else out += (char)ch;
    }
}
static void help_sketch() {
    fprintf(stderr,
            "rkmh sketch -f <seqs.fa|fq> [-k <k>]... [-s <sketch>] [-o <out.json>] [--kmer-cache <file>]\n"
            "  writes a JSON array with one MinHash sketch per sequence (schema of the reference's dump_hash_json);\n"
            "  `rkmh stream -R <out.json>` loads it instead of sketching references again;\n"
            "  --kmer-cache <file>: also enumerates the k-mers behind these sketches (k 8 .. 18) into <file>, which\n"
            "  `rkmh stream -R <out.json> --kmer-cache <file>` then loads instead of enumerating them at every start\n"
            "  the file records the hashing policy (\"hashPolicy\"); stream -R refuses sketches hashed under another one\n" HASH_POLICY_HELP);
}
static int main_sketch(int argc, char** argv) {
    std::vector<const char*> files;
    std::vector<int> ks;
    int S = 1000, device = 0;
    const char* outp = nullptr;
    const char* kmer_cache = nullptr;
    if (argc <= 2) { help_sketch(); exit(1); }
    optind = 2;
    int c;
    static struct option long_options[] = {{"help", no_argument, 0, 'h'}, {"kmer", required_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'}, {"reference", required_argument, 0, 'r'}, {"sketch-size", required_argument, 0, 's'},
        {"output", required_argument, 0, 'o'}, {"device", required_argument, 0, 1000}, {"kmer-cache", required_argument, 0, 1003}, HASH_POLICY_OPTION, {0, 0, 0, 0}};
    while ((c = getopt_long(argc, argv, "hk:f:r:s:o:t:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 1003: kmer_cache = optarg; break;
            case 'f': case 'r': files.push_back(optarg); break;
            case 'k': ks.push_back(atoi(optarg)); break;
            case 's': S = atoi(optarg); break;
            case 'o': outp = optarg; break;
            case 't': break;
            case 1000: device = atoi(optarg); break;
            default: help_sketch(); exit(1);
        }
    }
    if (ks.empty()) { fprintf(stderr, "No kmer size(s) provided. Will use a default kmer size of 16.\n"); ks.push_back(16); }
    if (files.empty()) { fprintf(stderr, "rkmh: -f <file> is required\n"); exit(1); }
    rk_ctx* ctx = nullptr;
    CK(rk_ctx_create(device, &g_policy, &ctx));
    rk_seqset s;
    CK(rk_parse_files(files.data(), (int)files.size(), &s));
    std::vector<uint64_t> sk((size_t)s.nseq * (size_t)S);
    std::vector<int32_t> lens((size_t)s.nseq);
    CK(rk_sketch_batch(ctx, s.bases, s.offsets, s.nseq, ks.data(), (int)ks.size(), S, sk.data(), lens.data()));
    if (kmer_cache && *kmer_cache) {
        // the index of these sketches is built once here, for its k-mer enumeration: the file's tag hashes the index keys, k and the
        // hashing policy, so a later `stream -R <these sketches> --kmer-cache <file>` finds it -- and anything else does not use it
        CK(rk_set_kmer_cache(ctx, kmer_cache));
        CK(rk_set_reference_sketches(ctx, sk.data(), lens.data(), (int)s.nseq, ks.data(), (int)ks.size(), S));
        if (rk_kmer_cache_state(ctx) == 0) fprintf(stderr, "rkmh: no k-mer enumeration for these sketches (k-mer sizes outside 8 .. 18, or a hash with two k-mers): %s not written\n", kmer_cache);
    }
    FILE* fo = outp ? fopen(outp, "w") : stdout;
    if (!fo) { fprintf(stderr, "rkmh: cannot write %s\n", outp); exit(1); }
    std::string kstr;
    for (size_t i = 0; i < ks.size(); ++i) { kstr += std::to_string(ks[i]); if (i + 1 < ks.size()) kstr += ' '; }
    std::string o = "[";
    char num[32];
    // "hashPolicy": this build's addition to dump_hash_json's keys (src/rkmh.cpp:489-525) -- what hashType / hashSeed leave open
    const std::string pol_text = policy_text(g_policy);
    for (int64_t i = 0; i < s.nseq; ++i) {
        std::string name;
        json_escape(name, s.names + s.name_offsets[i]);
        if (i) o += ',';
        o += "{\"alphabet\":\"ATGC\",\"canonical\":\"true\",\"hashBits\":64,\"hashPolicy\":\"" + pol_text + "\",\"hashSeed\":" + std::to_string(g_policy.seed) +
             ",\"hashType\":\"MurmurHash3_x64_128\",\"kmer\":\"" + kstr +
             "\",\"name\":\"" + name + "\",\"preserveCase\":\"false\",\"seqLen\":" + std::to_string(s.offsets[i + 1] - s.offsets[i]) +
             ",\"sketches\":{\"comment\":\"\",\"hashes\":[";
        for (int j = 0; j < lens[(size_t)i]; ++j) {
            int n = snprintf(num, sizeof num, j ? ",%llu" : "%llu", (unsigned long long)sk[(size_t)i * S + j]);
            o.append(num, (size_t)n);
        }
        o += "],\"length\":" + std::to_string(S) + ",\"name\":\"" + name + "\"}}";
        if (o.size() > (1u << 22)) { fwrite(o.data(), 1, o.size(), fo); o.clear(); }
    }
    o += "]\n";
    fwrite(o.data(), 1, o.size(), fo);
    if (fo != stdout) fclose(fo);
    rk_seqset_free(&s);
    rk_ctx_destroy(ctx);
    return 0;
}

// minimal reader for the files written above (tolerates whitespace; no general JSON support is claimed)
static bool json_find(const std::string& t, size_t from, size_t to, const char* key, size_t& vpos) {
    std::string pat = std::string("\"") + key + "\"";
    size_t p = t.find(pat, from);
    if (p == std::string::npos || p >= to) return false;
    p = t.find(':', p + pat.size());
    if (p == std::string::npos || p >= to) return false;
    ++p;
    while (p < to && isspace((unsigned char)t[p])) ++p;
    vpos = p;
    return true;
}
static std::string json_string_at(const std::string& t, size_t p) {
    std::string r;
    if (t[p] != '"') return r;
    for (++p; p < t.size() && t[p] != '"'; ++p) {
        if (t[p] == '\\' && p + 1 < t.size()) { ++p; r += t[p]; } else r += t[p];
    }
    return r;
}
static bool load_sketch_json(const char* path, LoadedSketches& L) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    std::string t;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) t.append(buf, n);
    fclose(f);
    // objects are delimited by their "sketches":{...}} tail; walk by the "alphabet" key that opens each object
    size_t pos = 0;
    std::vector<std::vector<uint64_t>> all;
    while ((pos = t.find("\"alphabet\"", pos)) != std::string::npos) {
        size_t next = t.find("\"alphabet\"", pos + 10);
        size_t end = next == std::string::npos ? t.size() : next;
        size_t v;
        if (!json_find(t, pos, end, "kmer", v)) return false;
        std::vector<int> ks;
        { std::string kk = json_string_at(t, v); char* e = &kk[0]; while (*e) { while (*e == ' ') ++e; if (!*e) break; ks.push_back((int)strtol(e, &e, 10)); } }
        if (L.ks.empty()) L.ks = ks; else if (ks != L.ks) return false;
        { // the policy the sketches were hashed under (absent: a file of an earlier build, which knew the defaults only)
            std::string pol = "default";
            if (json_find(t, pos, end, "hashPolicy", v)) pol = json_string_at(t, v);
            if (L.names.empty()) L.policy = pol; else if (pol != L.policy) return false;
        }
        if (!json_find(t, pos, end, "name", v)) return false;
        L.names.push_back(json_string_at(t, v));
        size_t sp;
        if (!json_find(t, pos, end, "sketches", sp)) return false;
        if (!json_find(t, sp, end, "length", v)) return false;
        int S = (int)strtol(t.c_str() + v, nullptr, 10);
        if (L.S == 0) L.S = S; else if (S != L.S) return false;
        if (!json_find(t, sp, end, "hashes", v)) return false;
        std::vector<uint64_t> h;
        const char* q = t.c_str() + v;
        if (*q != '[') return false;
        ++q;
        for (;;) {
            while (*q && (isspace((unsigned char)*q) || *q == ',')) ++q;
            if (*q == ']' || !*q) break;
            char* e;
            h.push_back(strtoull(q, &e, 10));
            if (e == q) return false;
            q = e;
        }
        all.push_back(h);
        pos = end;
    }
    if (all.empty() || L.S <= 0) return false;
    L.sk.assign(all.size() * (size_t)L.S, 0);
    for (size_t i = 0; i < all.size(); ++i) {
        if ((int)all[i].size() > L.S) return false;
        L.lens.push_back((int32_t)all[i].size());
        for (size_t j = 0; j < all[i].size(); ++j) L.sk[i * (size_t)L.S + j] = all[i][j];
    }
    return true;
}

static int main_hash(int argc, char** argv) {
    std::vector<const char*> files;
    std::vector<int> ks;
    int device = 0;
    bool print_kmers = false;
    if (argc <= 2) { help_hash(); exit(1); }
    static struct option long_options[] = {
        {"help", no_argument, 0, 'h'},        {"kmer", required_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'}, {"sketch-size", required_argument, 0, 's'},
        {"threads", required_argument, 0, 't'}, {"min-kmer-occurence", required_argument, 0, 'M'},
        {"max-samples", required_argument, 0, 'I'}, {"output", required_argument, 0, 'o'},
        {"device", required_argument, 0, 1000}, HASH_POLICY_OPTION, {0, 0, 0, 0}};
    optind = 2;
    int c;
    bool use_freqs = false;
    while ((c = getopt_long(argc, argv, "ThcwKk:f:s:t:mM:I:o:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 'f': files.push_back(optarg); break;
            case 'k': ks.push_back(atoi(optarg)); break;
            case 'K': print_kmers = true; break;
            case 'M': case 'I': use_freqs = true; break;       // accepted; nothing is printed (rkmh.cpp:2047,2109-2111)
            case 'T': case 'c': case 'w': case 'm': case 's': case 't': case 'o': break; // accepted and ignored
            case 1000: device = atoi(optarg); break;
            case '?': case 'h': default: print_help(); exit(1);
        }
    }
    if (ks.empty()) {
        fprintf(stderr, "No kmer size(s) provided. Will use a default kmer size of 16.\n");
        ks.push_back(16);
    }
    if (files.empty()) { fprintf(stderr, "rkmh: -f <file> is required\n"); exit(1); }
    if (use_freqs) return 0;
    rk_ctx* ctx = nullptr;
    if (!print_kmers) CK(rk_ctx_create(device, &g_policy, &ctx));
    rk_reader* rd = nullptr;
    CK(rk_reader_open(files[0], &rd)); // only input_files[0] is used, rkmh.cpp:2064,2085
    std::string buf;
    for (;;) {
        rk_seqset s;
        CK(rk_reader_next(rd, 1000, 1ull << 28, &s)); // 1000-record buffers, rkmh.cpp:2085-2094
        if (s.nseq == 0) { rk_seqset_free(&s); break; }
        buf.clear();
        if (print_kmers) {
            for (int64_t i = 0; i < s.nseq; ++i) {
                buf += s.names + s.name_offsets[i];
                const uint8_t* seq = s.bases + s.offsets[i];
                int64_t len = (int64_t)(s.offsets[i + 1] - s.offsets[i]);
                for (int k : ks)
                    for (int64_t w = 0; w + k < len + (g_policy.drop_last_window ? 0 : 1); ++w) { // len-k windows, or len-k+1 (policy U3)
                        buf += '\t';
                        for (int j = 0; j < k; ++j) {
                            signed char ch = (signed char)seq[w + j];
                            buf += (char)(((int)ch - 91) > 0 ? ch - 32 : ch);
                        }
                    }
                buf += '\n';
            }
        } else {
            std::vector<uint64_t> ho((size_t)s.nseq + 1);
            uint64_t* h = nullptr;
            CK(rk_hash_batch(ctx, s.bases, s.offsets, s.nseq, ks.data(), (int)ks.size(), &h, ho.data()));
            char num[32];
            for (int64_t i = 0; i < s.nseq; ++i) {
                buf += s.names + s.name_offsets[i];
                for (uint64_t j = ho[(size_t)i]; j < ho[(size_t)i + 1]; ++j) {
                    int n = snprintf(num, sizeof num, "\t%llu", (unsigned long long)h[j]);
                    buf.append(num, (size_t)n);
                }
                buf += '\n';
            }
            rk_free(h);
        }
        fwrite(buf.data(), 1, buf.size(), stdout);
        rk_seqset_free(&s);
    }
    rk_reader_close(rd);
    if (ctx) rk_ctx_destroy(ctx);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// hpv16 (main_hpv16, src/rkmh.cpp:2366-2723): HPV type of every read (set intersection of ALL its k-mer hashes with all hashes
// of each type reference), and its similarity to the k-mers specific to each HPV16 lineage / sublineage.
// Two mkmh functions on this path are absent from the reference snapshot (hash_set_intersection_size :2673, sort_by_similarity
// :2688/:2700); what they are taken to do is stated in DESIGN.md (policies U13/U14) and restated in oracle/oracle.py::hpv16.
#include <algorithm>
#include <set>
static void help_hpv16() {
    fprintf(stderr,
            "rkmh hpv16 -f <reads.fq> [-R <dir>] [-k <k>]... [-t <n>] [-M <n>]\n"
            "  classifies every read to an HPV type (<dir>/all_pave_ref.fa) and reports its k-mer matches to the\n"
            "  HPV16 lineages / sublineages of <dir>/new_refs.fa; <dir> defaults to ./data (as the reference: run it from the\n"
            "  rkmh directory).  Also writes lineage_specific_hashes.<k>.tst into the working directory.\n"
            "  -s/-N/-D are accepted and unused, as in the reference.  --device <id>: GPU to use.\n"
            "  Reads of any length are accepted; those with more than 16384 k-mers (all -k together) are answered one at a time.\n" HASH_POLICY_HELP);
}
static int main_hpv16(int argc, char** argv) {
    std::vector<const char*> read_files;
    std::string refpath = "data";                       // :2369
    std::vector<int> ks;
    int min_kmer_occ = 0, device = 0;
    bool do_read_depth = false;
    if (argc <= 2) { help_hpv16(); exit(1); }           // :2386-2389 (prints the classify help there)
    static struct option long_options[] = {             // :2393-2405
        {"help", no_argument, 0, 'h'},           {"kmer", no_argument, 0, 'k'},
        {"fasta", required_argument, 0, 'f'},    {"reference", required_argument, 0, 'r'},
        {"sketch", required_argument, 0, 's'},   {"threads", required_argument, 0, 't'},
        {"min-kmer-occurence", required_argument, 0, 'M'}, {"min-matches", required_argument, 0, 'N'},
        {"min-diff", required_argument, 0, 'D'}, {"max-samples", required_argument, 0, 'I'},
        {"device", required_argument, 0, 1000},  HASH_POLICY_OPTION, {0, 0, 0, 0}};
    optind = 2;
    int c;
    while ((c = getopt_long(argc, argv, "hk:f:R:s:t:M:N:D:", long_options, nullptr)) != -1) {
        switch (c) {
            case 1004: policy_apply(optarg, "--hash-policy"); break;
            case 't': case 's': case 'N': case 'D': break;          // parsed; nothing downstream reads them (:2411, :2428, :2435-2440)
            case 'f': read_files.push_back(optarg); break;
            case 'R': refpath = optarg; break;
            case 'k': if (optarg) ks.push_back(atoi(optarg)); break; // (--kmer is declared no_argument there too: unusable)
            case 'M': min_kmer_occ = atoi(optarg); do_read_depth = true; break;
            case 1000: device = atoi(optarg); break;
            case '?': case 'h': print_help(); exit(1);
            default: print_help(); abort();                          // --reference / --max-samples: declared, no case (:2441-2443)
        }
    }
    if (ks.empty()) {
        fprintf(stderr, "NO KMER SIZE PROVIDED. USING A DEFAULT KMER SIZE OF 16\n");   // :2449
        ks.push_back(16);
    }
    auto existing = [](const std::string& p) -> std::string {      // bundled test data is kept gzipped
        FILE* f = fopen(p.c_str(), "rb");
        if (f) { fclose(f); return p; }
        f = fopen((p + ".gz").c_str(), "rb");
        if (f) { fclose(f); return p + ".gz"; }
        fprintf(stderr, "rkmh hpv16: cannot open %s (pass the directory holding all_pave_ref.fa and new_refs.fa with -R)\n", p.c_str());
        exit(1);
    };
    const std::string type_file = existing(refpath + "/all_pave_ref.fa"), sub_file = existing(refpath + "/new_refs.fa");   // :2453-2456
    double t0 = now_s();
    rk_ctx* ctx = nullptr;
    CK(rk_ctx_create(device, &g_policy, &ctx));
    const rk_policy pol = g_policy;
    rk_seqset types, subs, reads;
    const char* p1[1] = {type_file.c_str()};
    const char* p2[1] = {sub_file.c_str()};
    CK(rk_parse_files(p1, 1, &types));
    CK(rk_parse_files(p2, 1, &subs));
    if (types.nseq < 1 || subs.nseq < 1) { fprintf(stderr, "rkmh hpv16: no sequences in the reference files\n"); exit(1); }
    memset(&reads, 0, sizeof reads);
    if (!read_files.empty()) CK(rk_parse_files(read_files.data(), (int)read_files.size(), &reads));
    tick("parse", t0);
    // all hashes (first -k only, :2546 and :2553) of the type and of the lineage/sublineage references, on the GPU
    const int k0 = ks[0];
    uint64_t *th = nullptr, *sh = nullptr;
    std::vector<uint64_t> tho((size_t)types.nseq + 1), sho((size_t)subs.nseq + 1);
    CK(rk_hash_batch(ctx, types.bases, types.offsets, types.nseq, &k0, 1, &th, tho.data()));
    CK(rk_hash_batch(ctx, subs.bases, subs.offsets, subs.nseq, &k0, 1, &sh, sho.data()));
    // lineage- and sublineage-specific k-mers: union per (sub)lineage, minus every other one (:2560-2650), in std::map order
    auto specific = [&](int key_len, std::vector<std::string>& names, std::vector<std::vector<uint64_t>>& lists) {
        std::map<std::string, std::set<uint64_t>> groups;
        for (int64_t i = 0; i < subs.nseq; ++i) {
            std::string key(subs.names + subs.name_offsets[i]);
            key = key.substr(0, (size_t)key_len);                    // subtype_keys[i][0] / substr(0, 2)
            groups[key].insert(sh + sho[(size_t)i], sh + sho[(size_t)i + 1]);
        }
        for (auto& x : groups) {
            std::vector<uint64_t> xdiff(x.second.begin(), x.second.end()), diff;
            for (auto& y : groups) {
                if (y.first == x.first) continue;
                diff.clear();
                std::set_difference(xdiff.begin(), xdiff.end(), y.second.begin(), y.second.end(), std::back_inserter(diff));
                xdiff.swap(diff);
            }
            names.push_back(x.first);
            lists.push_back(xdiff);                                  // ascending (mkmh::sort of a sorted range, :2592)
        }
    };
    std::vector<std::string> lin_names, sublin_names;
    std::vector<std::vector<uint64_t>> lin_lists, sublin_lists;
    specific(1, lin_names, lin_lists);
    {   // :2598-2611
        FILE* ofi = fopen(("lineage_specific_hashes." + std::to_string(k0) + ".tst").c_str(), "w");
        fprintf(stderr, "Lineage specific kmer table created:\n");
        for (size_t i = 0; i < lin_names.size(); ++i) {
            fprintf(stderr, "\t%s\t%zu\n", lin_names[i].c_str(), lin_lists[i].size());
            if (ofi) {
                fprintf(ofi, "%s\t", lin_names[i].c_str());
                for (uint64_t x : lin_lists[i]) fprintf(ofi, "%llu\t", (unsigned long long)x);
                fprintf(ofi, "\n");
            }
        }
        if (ofi) fclose(ofi);
    }
    specific(2, sublin_names, sublin_lists);
    fprintf(stderr, "Sublineage specific kmer table created:\n");   // :2647-2650
    for (size_t i = 0; i < sublin_names.size(); ++i) fprintf(stderr, "\t%s\t%zu\n", sublin_names[i].c_str(), sublin_lists[i].size());
    // reference lists for the device: distinct non-zero values, ascending (set semantics of hash_set_intersection_size, U13)
    const int ntype = (int)types.nseq, nlin = (int)lin_names.size(), nsub = (int)sublin_names.size(), nref = ntype + nlin + nsub;
    const int S = RK_MAX_SKETCH;   // list capacity = most hashes a read may have here (no bottom-s on this path)
    std::vector<uint64_t> lists((size_t)nref * (size_t)S, 0);
    std::vector<int32_t> lens((size_t)nref, 0);
    std::vector<size_t> full_len((size_t)nref, 0);                   // reflens as the reference passes them to sort_by_similarity
    auto put = [&](int r, std::vector<uint64_t> v) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        if (!v.empty() && v[0] == 0) v.erase(v.begin());
        if (v.size() > (size_t)S) { fprintf(stderr, "rkmh hpv16: reference %d has %zu distinct k-mers (limit %d)\n", r, v.size(), S); exit(1); }
        memcpy(&lists[(size_t)r * S], v.data(), v.size() * 8);
        lens[(size_t)r] = (int32_t)v.size();
    };
    for (int i = 0; i < ntype; ++i) put(i, std::vector<uint64_t>(th + tho[(size_t)i], th + tho[(size_t)i + 1]));
    for (int i = 0; i < nlin; ++i) { put(ntype + i, lin_lists[(size_t)i]); full_len[(size_t)(ntype + i)] = lin_lists[(size_t)i].size(); }
    for (int i = 0; i < nsub; ++i) { put(ntype + nlin + i, sublin_lists[(size_t)i]); full_len[(size_t)(ntype + nlin + i)] = sublin_lists[(size_t)i].size(); }
    rk_free(th); rk_free(sh);
    CK(rk_set_kmer_form(ctx, 0));   // these "references" are only used through the general kernels: no need to enumerate the k-mer universe
    CK(rk_set_reference_sketches(ctx, lists.data(), lens.data(), nref, ks.data(), (int)ks.size(), S));   // reads are hashed with EVERY -k (:2661)
    tick("tables", t0);
    rk_counter* cnt = nullptr;
    if (do_read_depth) {                                             // :2514-2530, then mask_by_frequency per read (:2663)
        CK(rk_counter_create(ctx, 800000000ull, &cnt));
        CK(rk_count_batch(ctx, reads.bases, reads.offsets, reads.nseq, cnt));
        CK(rk_set_depth_filter(ctx, cnt, min_kmer_occ));
    }
    std::vector<int32_t> out4((size_t)reads.nseq * 4), tail((size_t)reads.nseq * (size_t)(nlin + nsub));
    // The batched path keeps every hash of a read in the in-LDS sorter (RK_MAX_SKETCH values).  A longer read (a nanopore or
    // rolling-circle read of more than ~16 kb, or ~8 kb with two -k) is answered one at a time instead: hashed on the GPU
    // (rk_hash_batch, any length), masked (-M), then intersected with every list on the host exactly as :2666-2704 does -- the
    // reference handles reads of any length, so does this.
    auto hashes_of = [&](int64_t i) -> int64_t {
        const int64_t len = (int64_t)(reads.offsets[i + 1] - reads.offsets[i]);
        int64_t hn = 0;
        for (int k : ks) { const int64_t nw = pol.drop_last_window ? len - k : len - k + 1; if (nw > 0) hn += nw; }
        return hn;
    };
    std::vector<int64_t> longs, normal;
    for (int64_t i = 0; i < reads.nseq; ++i) (hashes_of(i) > (int64_t)S ? longs : normal).push_back(i);
    if (longs.empty()) {
        if (reads.nseq > 0) CK(rk_classify_groups_batch(ctx, reads.bases, reads.offsets, reads.nseq, ntype, out4.data(), tail.data()));
    } else {
        fprintf(stderr, "rkmh hpv16: %zu read(s) with more than %d k-mers are classified one at a time\n", longs.size(), S);
        if (!normal.empty()) { // the other reads as a batch of their own
            std::vector<uint64_t> off(normal.size() + 1, 0);
            for (size_t j = 0; j < normal.size(); ++j) off[j + 1] = off[j] + (reads.offsets[normal[j] + 1] - reads.offsets[normal[j]]);
            std::vector<uint8_t> sub((size_t)off.back() + 64);
            for (size_t j = 0; j < normal.size(); ++j) memcpy(sub.data() + off[j], reads.bases + reads.offsets[normal[j]], (size_t)(off[j + 1] - off[j]));
            std::vector<int32_t> o4(normal.size() * 4), tl(normal.size() * (size_t)(nlin + nsub));
            CK(rk_classify_groups_batch(ctx, sub.data(), off.data(), (int64_t)normal.size(), ntype, o4.data(), tl.data()));
            for (size_t j = 0; j < normal.size(); ++j) {
                memcpy(&out4[(size_t)normal[j] * 4], &o4[j * 4], 16);
                memcpy(&tail[(size_t)normal[j] * (size_t)(nlin + nsub)], &tl[j * (size_t)(nlin + nsub)], sizeof(int32_t) * (size_t)(nlin + nsub));
            }
        }
        for (int64_t i : longs) {
            uint64_t* h = nullptr;
            uint64_t ho[2] = {0, 0};
            const uint64_t one[2] = {0, reads.offsets[i + 1] - reads.offsets[i]};
            CK(rk_hash_batch(ctx, reads.bases + reads.offsets[i], one, 1, ks.data(), (int)ks.size(), &h, ho));
            if (cnt) CK(rk_mask_by_frequency(ctx, h, (int)ho[1], cnt, min_kmer_occ));
            std::vector<uint64_t> v(h, h + ho[1]);
            rk_free(h);
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            if (!v.empty() && v[0] == 0) v.erase(v.begin());
            auto isect = [&](int r) { // distinct non-zero values in both ascending arrays (U13)
                const uint64_t* a = &lists[(size_t)r * S];
                const int na = lens[(size_t)r];
                int n = 0, x = 0; size_t y = 0;
                while (x < na && y < v.size()) { if (a[x] == v[y]) { ++n; ++x; ++y; } else if (a[x] < v[y]) ++x; else ++y; }
                return n;
            };
            int best = 0, best_id = 0, prev = -1;                                   // first maximum wins (:2669-2679)
            for (int r = 0; r < ntype; ++r) { const int c2 = isect(r); if (c2 > best) { prev = best; best = c2; best_id = r; } }
            int32_t* o = &out4[(size_t)i * 4];
            o[0] = best_id; o[1] = best; o[2] = best - prev; o[3] = (int32_t)v.size();
            for (int r = 0; r < nlin + nsub; ++r) tail[(size_t)i * (size_t)(nlin + nsub) + (size_t)r] = isect(ntype + r);
        }
    }
    tick("classify", t0);
    const bool den_read = getenv("RKMH_HPV16_SIM") && !strcmp(getenv("RKMH_HPV16_SIM"), "read");   // U14: similarity denominator
    // The lines (one stable sort and a dozen "%g" per read) are written by all granted CPUs, 16 k reads per piece, and leave in input order.
    auto emit_range = [&](int64_t lo, int64_t hi, std::string& buf) {
        char num[64];
        std::vector<int> order;
        std::vector<double> sims;
        auto ranked = [&](const int32_t* cnts, int first, int n, int hashnum, const std::vector<std::string>& names, std::string& a, std::string& b) {
            // sort_by_similarity (U14): intersection / list size, descending, ties in reference order
            order.resize((size_t)n); sims.resize((size_t)n);
            for (int i = 0; i < n; ++i) {
                order[(size_t)i] = i;
                const double den = den_read ? (double)hashnum : (double)full_len[(size_t)(first + i)];
                sims[(size_t)i] = den > 0 ? (double)cnts[i] / den : 0.0;
            }
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sims[(size_t)x] > sims[(size_t)y]; });
            for (int i : order) {
                a += names[(size_t)i]; a += ':';
                snprintf(num, sizeof num, "%g", sims[(size_t)i]);        // ostream << double
                a += num; a += ';';
                b += std::to_string(cnts[i]); b += ';';
            }
        };
        std::string la, lb, sa, sb;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t len = (int64_t)(reads.offsets[i + 1] - reads.offsets[i]);
            int64_t hashnum = 0;
            for (int k : ks) { const int64_t nw = pol.drop_last_window ? len - k : len - k + 1; if (nw > 0) hashnum += nw; }
            const int32_t* r = &out4[(size_t)i * 4];
            const int32_t* t = &tail[(size_t)i * (size_t)(nlin + nsub)];
            buf += reads.names + reads.name_offsets[i]; buf += '\t';
            buf += types.names + types.name_offsets[r[0]]; buf += '\t';
            buf += std::to_string(r[1]); buf += '/'; buf += std::to_string(hashnum); buf += '\t';
            la.clear(); lb.clear(); sa.clear(); sb.clear();
            ranked(t, ntype, nlin, (int)hashnum, lin_names, la, lb);
            ranked(t + nlin, ntype + nlin, nsub, (int)hashnum, sublin_names, sa, sb);
            buf += la; buf += '\t'; buf += sa; buf += '\t'; buf += lb; buf += '\t'; buf += sb; buf += '\n';
        }
    };
    {
        const int64_t PIECE = 1 << 14;
        const int nth = std::max(1, std::min(granted_cpus_main(), 32));
        const int64_t npieces = (reads.nseq + PIECE - 1) / PIECE;
        for (int64_t p0 = 0; p0 < npieces; p0 += nth) { // a wave of pieces at a time: memory stays bounded, the order is the input's
            const int64_t np = std::min<int64_t>(nth, npieces - p0);
            std::vector<std::string> bufs((size_t)np);
            std::vector<std::thread> th;
            for (int64_t q = 0; q < np; ++q)
                th.emplace_back([&, q] { emit_range((p0 + q) * PIECE, std::min(reads.nseq, (p0 + q + 1) * PIECE), bufs[(size_t)q]); });
            for (auto& t : th) t.join();
            for (auto& b : bufs) fwrite(b.data(), 1, b.size(), stdout);
        }
    }
    tick("emit", t0);
    if (cnt) rk_counter_destroy(cnt);
    rk_seqset_free(&types); rk_seqset_free(&subs);
    if (!read_files.empty()) rk_seqset_free(&reads);
    rk_ctx_destroy(ctx);
    return 0;
}


static pid_t g_child = -1;
static void forward_signal(int sig) { if (g_child > 0) kill(g_child, sig); }
// see tell_parent: the parent's side.  Returns in the child (and in a process that does not fork); the parent never returns.
static void fork_for_fast_exit() {
    if (getenv("RKMH_SLOW_EXIT") || (getenv("RKMH_FORK") && atoi(getenv("RKMH_FORK")) == 0) || under_profiler()) return;
    int fds[2];
    if (pipe(fds) != 0) return;
    fflush(stdout); fflush(stderr);
    const pid_t pid = fork();
    if (pid < 0) { close(fds[0]); close(fds[1]); return; }
    if (pid == 0) { close(fds[0]); g_done_fd = fds[1]; return; }
    close(fds[1]);
    close(1); // (the parent writes nothing: the child's descriptor is the file's last one -- see tell_parent)
    g_child = pid;
    for (int sig : {SIGINT, SIGTERM, SIGHUP, SIGQUIT, SIGABRT, SIGPIPE}) signal(sig, forward_signal); // (timeout(1), ^C: they mean the worker)
    close(0); // (the child reads standard input, if anyone does)
    unsigned char b = 0;
    ssize_t n;
    while ((n = read(fds[0], &b, 1)) < 0 && errno == EINTR) {}
    if (n == 1) { // the output is complete: the child finishes dying on its own
        if (getenv("RKMH_TIMING")) {
            char line[96];
            const int len = snprintf(line, sizeof line, "[rkmh timing] parent released at epoch %.3f\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
            if (len > 0 && write(2, line, (size_t)len) < 0) {}
        }
        _exit((int)b);
    }
    int st = 0;
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
    if (WIFEXITED(st)) _exit(WEXITSTATUS(st));
    if (WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); raise(WTERMSIG(st)); _exit(128 + WTERMSIG(st)); }
    _exit(1);
}

int main(int argc, char** argv) {
    if (argc <= 1) { print_help(); exit(1); }
    fork_for_fast_exit();
    // The device front end's workers have a stream each; the runtime maps streams onto four hardware queues unless told otherwise,
    // and kernels of two streams on one queue run one after the other -- the long inflate kernels of BGZF jobs above all
    // (profiles/r05_gz.txt: 2.4 of 8 launches overlapped, 4.2 with 16 queues).  Read by the runtime when it starts: set before any HIP call.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    rk_default_policy(&g_policy);
    if (const char* e = getenv("RKMH_POLICY")) policy_apply(e, "RKMH_POLICY");
    std::string cmd = argv[1];
    if (cmd == "stream") return main_stream(argc, argv);
    if (cmd == "classify") {
        fprintf(stderr, "CLASSIFY COMMAND IS TEMPORARILY UNAVAILABLE: TRY rkmh stream INSTEAD.\n"); // rkmh.cpp:2746
        return main_stream(argc, argv);
    }
    if (cmd == "hash") return main_hash(argc, argv);
    if (cmd == "filter") return main_filter(argc, argv);
    if (cmd == "call") return main_call(argc, argv);
    if (cmd == "sketch") return main_sketch(argc, argv);
    if (cmd == "pack") return main_pack(argc, argv);
    if (cmd == "hpv16") return main_hpv16(argc, argv);
    print_help();
    exit(1);
}
