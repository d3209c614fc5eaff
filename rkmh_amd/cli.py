"""`python -m rkmh_amd.cli stream|classify|filter ...` -- the multi-GPU form of the rkmh stream/classify/filter commands.

Same flags and stdout as main_stream (/root/reference/src/rkmh.cpp:584-989; option table :626-650; line
format :892).  Launched plainly it uses one GPU; launched under torch.distributed.run it is one process per
GPU: rank 0 sketches the references and broadcasts the sketches over RCCL, every rank classifies a contiguous
block of the reads, the -M path all-reduces the k-mer counter between its two passes
(rkmh.cpp:904-948), and rank 0 prints the lines in input order.  `filter` (main_filter, rkmh.cpp:996-1424, file mode)
shards the same way: 10 M-slot counters (:1187-1188), the decision of classify_and_count_diff_filter (equiv.hpp:324-353)
on the gathered rows, passing reads printed by rank 0 in input order.  (The single-GPU C++ binary is bin/rkmh; `filter -i`,
`call` and `hash` exist only there: a 52 k-read `call` takes 0.2 s on one GPU.)
"""
import getopt
import ctypes as C
import os
import sys

import numpy as np

from . import api, dist as rdist

HELP = """rkmh stream|classify -r <refs.fa> -f <reads.fq> [-k <k>]... [-s <sketch>] [-M n] [-I n] [-N n] [-D n] [--hash-policy <spec>]
  --hash-policy <spec>   presets (default, mash) and/or fold=swap32|h1|w2w1, windows=len-k|len-k+1, zero=count|skip, mask=lt|le,
                         freqmax=incl|excl, seed=<n> (rk_policy_parse); RKMH_POLICY: the same, read first
"""


def _upper(a):
    """to_upper as parse_fastas applies it (rkmh.cpp:280): signed chars above 91 lose 32."""
    x = np.ascontiguousarray(a, dtype=np.uint8).view(np.int8)
    return np.where(x > 91, x - 32, x).astype(np.int8).view(np.uint8).tobytes()


def _cut_before(fd, size, p):
    """The last record start (four-line rule, rk_fastq_cut) before byte p of the file: how byte ranges become ranges of whole records.
    Every rank and every block computes its ends with this one function, so neighbours agree without talking."""
    if p <= 0:
        return 0
    if p >= size:
        return size
    w = 1 << 16
    while True:
        lo = max(0, p - w)
        cut = api.fastq_cut(os.pread(fd, p - lo, lo))
        if cut > 0:
            return lo + cut
        if lo == 0:
            return 0
        w *= 8


def _raw_eligible(path):
    try:
        if not os.path.isfile(path) or os.path.getsize(path) == 0:
            return False
        with open(path, "rb") as f:
            return f.read(1) == b"@"
    except OSError:
        return False


class _RankOutput:
    """Standard output shared by the ranks (torch.distributed.run hands every worker the launcher's descriptor 1).  No text travels
    between the processes: into a regular file every rank writes its blocks at their final offsets, all ranks at once (the byte
    counts are all-gathered); into a pipe the ranks write one after the other.  A single rank streams its blocks out as they finish."""

    def __init__(self, fd, rank, world):
        import fcntl
        import socket
        import stat
        import zlib
        self.fd, self.rank, self.world = fd, rank, world
        st = os.fstat(fd)
        self.regular = stat.S_ISREG(st.st_mode) and not (fcntl.fcntl(fd, fcntl.F_GETFL) & os.O_APPEND)
        self.base = os.lseek(fd, 0, os.SEEK_CUR) if self.regular else 0
        # Writing at final offsets (or in turn into a pipe) is only right when every rank's descriptor is the SAME open file: one
        # host, one device and inode, the same kind and the same offset.  A launcher that gives each rank its own stdout
        # (torchrun --redirects / --log-dir) or ranks on several hosts break that -- every rank would write its fragment at a global
        # offset of ITS file.  Then rank 0 gathers the text and writes it alone (`shared` False).
        ident = (zlib.crc32(socket.gethostname().encode()), int(st.st_dev), int(st.st_ino), 1 if self.regular else 0, int(self.base))
        self.shared = True
        if world > 1:
            for v in ident:
                got = rdist.all_gather_int(v)
                self.shared = self.shared and all(x == got[0] for x in got)

    def _put(self, piece, at):
        mv, done = memoryview(piece), 0
        while done < len(mv):
            done += os.pwrite(self.fd, mv[done:], at + done) if self.regular else os.write(self.fd, mv[done:])

    def stream(self, piece):
        """world == 1: the next block of the output, in order."""
        self._put(piece, self.base)
        self.base += len(piece)

    def pieces_of_a_file(self, pieces):
        """This rank's output for ONE input file (a list of blocks, in order); returns when every rank's part is written."""
        if not self.shared: # the ranks' descriptors are different files: all text to rank 0, which writes it in rank order
            text = rdist.gather_bytes(b"".join(bytes(p) for p in pieces), dst=0)
            if self.rank == 0:
                self._put(text, self.base)
                self.base += len(text)
            return
        n = sum(len(p) for p in pieces)
        sizes = rdist.all_gather_int(n)
        if self.regular:
            at = self.base + sum(sizes[: self.rank])
            for p in pieces:
                self._put(p, at)
                at += len(p)
            self.base += sum(sizes)
            rdist.barrier()
        else:
            for r in range(self.world):
                if r == self.rank:
                    for p in pieces:
                        self._put(p, 0)
                rdist.barrier()

    def finish(self):
        if self.regular and self.rank == 0:
            os.lseek(self.fd, self.base, os.SEEK_SET)   # whatever is written next continues behind the blocks


def _references_on_device(ctx, refs, ks, sketch, max_samples, counter_slots, count_distinct):
    """The -r files as raw text through the device (rk_fasta_load_*): a few threads read blocks into page-locked buffers and upload
    them, the GPU strips header lines and line ends, the sketches are made from the packed bases where they lie.  For plain FASTA
    of 64 MB or more (RKMH_RAW_REFS=1: any size, =0: never); returns the reference names, or None: parse on the host."""
    import threading
    env = os.environ.get("RKMH_RAW_REFS")
    if env == "0":
        return None
    try:
        sizes = [os.path.getsize(p) for p in refs]
        for p in refs:
            with open(p, "rb") as f:
                if not os.path.isfile(p) or f.read(1) != b">":
                    return None
    except OSError:
        return None
    total = sum(sz + 1 for sz in sizes)          # a newline after every file
    if env != "1" and total < (64 << 20):
        return None
    block = max(4096, int(os.environ.get("RKMH_RAW_BLOCK_KB", "16384")) << 10)
    jobs, at = [], 0
    for i, sz in enumerate(sizes):
        for lo in range(0, sz, block):
            hi = min(sz, lo + block)
            jobs.append((i, lo, hi, at + lo, hi == sz))
        at += sz + 1
    fds = [os.open(p, os.O_RDONLY) for p in refs]
    nw = max(1, min(6, len(jobs)))
    slots, load = [], None
    state, lock, nxt = {"ok": True}, threading.Lock(), [0]
    try:
        load = api.FastaLoad(ctx, total)
        for _ in range(nw):
            slots.append(api.FastqSlot(ctx, max_bytes=block + 64))

        def work(slot):
            try:
                mv = memoryview(slot.text_buffer()).cast("B")
                while state["ok"]:
                    with lock:
                        j = nxt[0]
                        nxt[0] += 1
                    if j >= len(jobs):
                        return
                    i, lo, hi, at_, last = jobs[j]
                    n, got = hi - lo, 0
                    while got < n:
                        k = os.preadv(fds[i], [mv[got:n]], lo + got)
                        if k <= 0:
                            raise OSError("short read on %s" % refs[i])
                        got += k
                    if last:
                        mv[n] = 10
                        n += 1
                    load.put_raw(slot, at_, n)
            except Exception as e:      # (an exception would only end this thread: the text would be incomplete and nobody would know)
                state["err"] = e
                state["ok"] = False

        th = [threading.Thread(target=work, args=(sl,)) for sl in slots]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if not state["ok"]:
            if "err" in state:
                sys.stderr.write("rkmh: references through the device: %s; parsing on the host\n" % state["err"])
            return None
        st, names, _ = load.finish(total)
        if st != 0:
            return None
        if count_distinct:
            api._chk(load._lib.rk_set_reference_count_mode(ctx._h, 1))
        load.set_references(ks, sketch, max_samples=-1 if max_samples is None else int(max_samples), counter_slots=counter_slots)
        if os.environ.get("RKMH_TIMING"):
            sys.stderr.write("[rkmh timing] references through the device: %d sequences, %.0f MB of text\n" % (len(names), total / 1e6))
        return names
    except api.RkmhError as e:
        sys.stderr.write("rkmh: references through the device: %s; parsing on the host\n" % e)
        return None
    finally:
        for sl in slots:
            sl.destroy()
        if load is not None:
            load.destroy()
        for fd in fds:
            os.close(fd)


def _device_ingest(ctx, rank, local, world, reads, ref_names, sketch, min_occ, min_matches, min_diff, filter_mode, out_fd, compact_ok=False):
    """This rank's byte range of every read file through the device FASTQ front end (rk_fastq_slot_*): worker threads read raw blocks
    straight into page-locked buffers, the GPU splits / checks / packs / classifies them, the lines are written in C from the names
    where they lie (rk_fastq_stream_lines / rk_fastq_filter_records) -- the host never parses a read, exactly as bin/rkmh does it
    (rkmh_main.cpp, stream_file_raw).  -M: two such passes with the RCCL all-reduce of the table in between.  Returns this rank's
    output is written to out_fd (see _RankOutput) and True returned -- or, when ANY rank met text that is not four lines per record,
    nothing is written and False returned (then every rank takes the parsing path)."""
    import threading
    # BGZF (bgzip) files: chains of independent gzip members -- every rank takes its share of the members and its worker threads
    # inflate them (rk_bgzf_*), so compressed reads are neither inflated by one thread nor parsed by every rank
    bzs = [api.Bgzf.open(p) if (os.path.isfile(p) and os.environ.get("RKMH_BGZF", "1") != "0") else None for p in reads]
    ok = bool(reads) and all((bz.first_byte() == 0x40) if bz is not None else _raw_eligible(p) for p, bz in zip(reads, bzs))
    if not rdist.all_true(ok):
        for bz in bzs:
            if bz is not None:
                bz.close()
        return False
    block = max(4096, int(os.environ.get("RKMH_RAW_BLOCK_KB", "16384")) << 10)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    nw = int(os.environ.get("RKMH_RAW_WORKERS", "0")) or max(2, min(6, ((os.cpu_count() or 8) * 3 // 8) // local_world))
    if any(bz is not None for bz in bzs) and not os.environ.get("RKMH_RAW_WORKERS") and os.environ.get("RKMH_BGZF_DEVICE", "0") in ("", "0"):
        nw = max(nw, min(32, ((os.cpu_count() or 8) - 2) // local_world))      # inflating on the host is CPU work: all but two of the CPUs
    fds = [os.open(p, os.O_RDONLY) for p in reads]
    sizes = [bz.members if bz is not None else os.fstat(fd).st_size for fd, bz in zip(fds, bzs)]
    slots, state = [], {"ok": True}
    lock = threading.Lock()
    try:
        # this rank's blocks of every file: [cut(size r / world), cut(size (r + 1) / world)) in steps of `block`, ends cut the same way
        plan = []
        for fd, size, bz in zip(fds, sizes, bzs):
            if bz is not None:      # jobs = runs of members holding about a block of text; this rank's share of them
                first = bz.plan(block - (256 << 10) if block > (1 << 20) else block * 3 // 4)
                nj = len(first) - 1
                plan.append([(first[j], first[j + 1]) for j in range(nj * rank // world, nj * (rank + 1) // world)])
                continue
            a, b = _cut_before(fd, size, size * rank // world), _cut_before(fd, size, size * (rank + 1) // world)
            ends = [_cut_before(fd, size, p) for p in range(a + block, b, block)] + [b]
            blocks, s0 = [], a
            for e in ends:
                if e > s0:
                    blocks.append((s0, e))
                    s0 = e
            plan.append(blocks)
        longest = max([e - s0 for blocks, bz in zip(plan, bzs) if bz is None for s0, e in blocks] + [4096] + ([block] if any(bz is not None for bz in bzs) else []))
        for _ in range(nw):
            slots.append(api.FastqSlot(ctx, max_bytes=longest + 64))
        parts = None if filter_mode else api.LineParts(ref_names, sketch, min_matches, min_diff)

        def run_pass(counter, sink=None):
            texts = []
            for fd, size, blocks, bz in zip(fds, sizes, plan, bzs):
                results, nxt, due = [None] * len(blocks), [0], [0]

                def work(slot, fd=fd, size=size, blocks=blocks, bz=bz, results=results, nxt=nxt, due=due):
                    try:
                        buf = slot.text_buffer()
                        mv = memoryview(buf).cast("B")
                        while state["ok"]:
                            with lock:
                                i = nxt[0]
                                nxt[0] += 1
                            if i >= len(blocks):
                                return
                            s0, s1 = blocks[i]
                            on_device = False
                            if bz is not None:      # members [s0, s1): inflated on the device (rk_inflate.hip), else here; cut to the whole records that start in them
                                st = 1
                                if os.environ.get("RKMH_BGZF_DEVICE", "0") not in ("", "0"):      # opt-in here: it pays with ~1 000 members per job and several jobs in flight (RKMH_RAW_BLOCK_KB=65536; bin/rkmh arranges that itself, profiles/r05_gz.txt)
                                    st, n, _ = slot.load_bgzf(bz, s0, s1)
                                    on_device = st == 0
                                if st != 0:
                                    st, n, _ = bz.fastq_records(s0, s1, C.addressof(buf), slot.max_bytes + 63)
                                if st != 0:
                                    state["ok"] = False
                                    return
                            else:
                                n, got = s1 - s0, 0
                                while got < n:
                                    k = os.preadv(fd, [mv[got:n]], s0 + got)
                                    if k <= 0:
                                        state["ok"] = False
                                        return
                                    got += k
                            if not on_device and s1 == size and n and mv[n - 1] != 10:
                                mv[n] = 10          # a last line without its newline (the slot holds spare bytes)
                                n += 1
                            if counter is not None:
                                st, _ = slot.count_raw(n, counter)
                                if st != 0:
                                    state["ok"] = False
                            else:
                                res = slot.classify_raw(n)
                                if res.status != 0:
                                    state["ok"] = False
                                    return
                                text = b"" if not res.nrec else (slot.filter_records(res, min_matches, min_diff) if filter_mode else slot.stream_lines(parts, res))
                                with lock:
                                    results[i] = text
                                    if sink is not None:      # one rank: blocks leave in order as soon as they are due
                                        while due[0] < len(blocks) and results[due[0]] is not None:
                                            sink.stream(results[due[0]])
                                            results[due[0]] = b""
                                            due[0] += 1
                    except Exception as e:      # (an exception would only end this thread and its block would be missing from the output)
                        state["err"] = e
                        state["ok"] = False

                th = [threading.Thread(target=work, args=(sl,)) for sl in slots]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                if "err" in state:
                    raise state["err"]
                texts.append([r for r in results if r])
            return texts

        counter = t = None
        if min_occ is not None:
            import torch
            nslots = 10000000 if filter_mode else 200000000  # rkmh.cpp:1187 / :739

            def depth_map(compact):
                # compact: only the slots of index keys (rk_counter_create_compact) -- the all-reduce then moves a few hundred KB
                tt = torch.zeros(api.Counter.compact_entries(ctx, nslots) if compact else nslots, dtype=torch.int32, device="cuda:%d" % local)
                torch.cuda.synchronize()
                return tt, api.Counter(ctx, slots=nslots, device_ptr=tt.data_ptr(), compact=compact)

            t, counter = depth_map(compact_ok)
            need_full = False
            try:
                run_pass(counter)                               # pass 1 (rkmh.cpp:904-910) on this rank's blocks
            except api.NeedFullDepthMap:                        # a read with more hashes than the sketch keeps
                need_full = True
                state.pop("err", None)
                state["ok"] = True
            ctx.synchronize()
            if compact_ok and not rdist.all_true(not need_full):   # on any rank: every rank repeats the pass into the full table
                counter.destroy()
                t, counter = depth_map(False)
                run_pass(counter)
                ctx.synchronize()
            if not rdist.all_true(state["ok"]):
                counter.destroy()
                return False
            rdist.allreduce_counter(t)                          # RCCL sum over ranks
            torch.cuda.synchronize()
            ctx.set_depth_filter(counter, min_occ)
        # One rank streams its blocks out while later ones are still on the device.  That commits output before the whole input is
        # known to be regular, so it is done only where that is harmless: after the counting pass of -M (which saw every block), or
        # into a regular file (cut back to where it was if a block is refused after all).  Into a pipe a single rank collects like
        # the others.  (bin/rkmh hands over to its scanner mid-file instead; this front end has no scanner thread.)
        sink = _RankOutput(out_fd, rank, world)
        base0 = sink.base
        streaming = world == 1 and (counter is not None or sink.regular)
        texts = run_pass(None, sink if streaming else None)
        good = rdist.all_true(state["ok"])
        if streaming and not good and sink.regular:
            os.ftruncate(out_fd, base0)
        elif streaming and not good:
            # blocks already left through a pipe and cannot be taken back: falling back to the parsing path would print them twice
            raise SystemExit("rkmh: %s changed between the two passes (or could not be read)" % ", ".join(reads))
        if os.environ.get("RKMH_TIMING"):
            sys.stderr.write("[rkmh timing] rank %d: device front end: %d blocks of %d file(s), %d worker threads%s\n"
                             % (rank, sum(len(b) for b in plan), len(plan), nw, "" if good else " -- refused, parsing on the host"))
        if counter is not None:
            ctx.set_depth_filter(None, 0)
            counter.destroy()
        if good and not streaming:
            for pieces in texts:          # file by file, every file's text in rank order
                sink.pieces_of_a_file(pieces)
        if good:
            sink.finish()
        return good
    finally:
        for sl in slots:
            sl.destroy()
        for fd in fds:
            os.close(fd)


def main_stream(argv, filter_mode=False):
    if len(argv) <= 2:
        sys.stderr.write(HELP)
        return 1
    longopts = ["help", "kmer=", "fasta=", "reference=", "sketch-size=", "ref-sketch=", "threads=", "min-kmer-occurence=",
                "min-matches=", "min-diff=", "max-samples=", "pre-reads=", "pre-references=", "read-kmer-map-file=",
                "ref-kmer-map-file=", "in-stream", "output-reads", "merge-sketch", "hash-policy="]
    try:
        opts, _ = getopt.getopt(argv[2:], "zmhdk:f:r:s:S:t:M:N:I:R:F:p:q:iD:", longopts)
    except getopt.GetoptError:
        sys.stderr.write(HELP)
        return 1
    refs, reads, ks = [], [], []
    sketch, min_occ, min_matches, min_diff, max_samples = 1000, None, -1, 0, None
    in_stream = False
    # the hashing policy: defaults, then RKMH_POLICY, then --hash-policy (as bin/rkmh; one parser for both: rk_policy_parse)
    try:
        policy = api.parse_policy(os.environ.get("RKMH_POLICY"))
        for o, a in opts:
            if o == "--hash-policy":
                policy = api.parse_policy(a, base=policy)
    except api.RkmhError as e:
        sys.stderr.write("rkmh: --hash-policy / RKMH_POLICY: %s\n" % e)
        return 1
    for o, a in opts:
        if o in ("-r", "--reference"): refs.append(a)
        elif o in ("-f", "--fasta"): reads.append(a)
        elif o in ("-k", "--kmer"): ks.append(int(a))
        elif o in ("-s", "--sketch-size"): sketch = int(a)
        elif o in ("-M", "--min-kmer-occurence"): min_occ = int(a)
        elif o in ("-I", "--max-samples"): max_samples = int(a)
        elif o in ("-N", "--min-matches"): min_matches = int(a)
        elif o in ("-D", "--min-diff"): min_diff = int(a)
        elif o in ("-i", "--in-stream"): in_stream = True
        elif o in ("-h", "--help", "-d"):
            sys.stderr.write(HELP)
            return 1
    if not ks:
        sys.stderr.write("No kmer size(s) provided. Will use a default kmer size of 16.\n")
        ks = [16]
    if not refs:
        sys.stderr.write("rkmh: at least one -r reference file is required\n")
        return 1
    if filter_mode and (in_stream or not reads):
        sys.stderr.write("rkmh_amd.cli filter: file mode only (-f); for -i use bin/rkmh filter\n")
        return 1
    # The result lines own standard output: descriptor 1 is kept aside for them and re-pointed at standard error for everything else
    # in the process (RCCL prints a version banner, gloo its connection messages -- to stdout, which a drop-in tool must not do).
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    out = os.fdopen(result_fd, "wb")
    rank, local, world = rdist.init()
    ctx = api.Context(local, policy_spec=api.describe_policy(policy))
    if os.environ.get("RKMH_KMER_CACHE"):
        ctx.set_kmer_cache(os.environ["RKMH_KMER_CACHE"])    # the k-mer enumeration of these references, kept between runs
    compact_ok = False
    if min_occ is not None and not (os.environ.get("RKMH_EXACT_MIN_NUM", "0") not in ("", "0")):
        # -M: the output compares num_mins with -N (stream, rkmh.cpp:938) or with 0 (filter, :1292) and nothing else, so
        # min(num_mins, bound) is all it needs -- the masked pass then looks up index keys, not every window (rk_set_min_num_bound)
        # (filter with -D >= 0: a read that shares nothing fails the diff test anyway, so not even min(read_min_lens, 1) is needed)
        cmp_with = (-1 if min_diff >= 0 else 0) if filter_mode else min_matches
        bound = 0 if cmp_with < 0 else cmp_with + 1
        ctx.set_min_num_bound(bound)
        compact_ok = bound == 0 and os.environ.get("RKMH_FULL_DEPTH_MAP", "0") in ("", "0")
    # Rank 0 alone reads the reference files -- genome-sized plain FASTA as raw text stripped on the device (_references_on_device),
    # anything else with the host parser --, sketches them and broadcasts sketches and names; the other ranks never open them.
    ms_filter = max_samples if (max_samples is not None and max_samples < 100000) else None   # filter: rkmh.cpp:1211
    names = None
    if rank == 0:
        names = _references_on_device(ctx, refs, ks, sketch, (ms_filter if filter_mode else max_samples), 10000000 if filter_mode else 0, filter_mode)
        if names is None:
            Rh = api.parse_files(refs)
            if Rh["nseq"] >= 1:
                if filter_mode:   # 10 M slots (:1188), filled once per distinct hash
                    ctx.set_references(Rh["bases"], Rh["offsets"], ks, sketch, max_samples=ms_filter, counter_slots=10000000, count_distinct=True)
                else:
                    ctx.set_references(Rh["bases"], Rh["offsets"], ks, sketch, max_samples=max_samples)
            names = Rh["names"]
            del Rh
        blob = b"\0".join(names) + b"\0" if names else b""
    else:
        blob = None
    blob = rdist.broadcast_bytes(blob, src=0)
    names = blob.split(b"\0")[:-1] if blob else []
    R = {"names": names, "nseq": len(names)}
    if R["nseq"] < 1:
        sys.stderr.write("rkmh: no reference sequences found\n")
        return 1
    if rank == 0:
        sk, ln = ctx.get_reference_sketches()
    else:
        sk = ln = None
    sk, ln = rdist.broadcast_sketches(sk, ln, R["nseq"], sketch, src=0)
    if rank != 0:
        ctx.set_reference_sketches(sk, ln, ks, sketch)
    # Uncompressed FASTQ files do not pass through a host parser at all (see _device_ingest); RKMH_RAW=0 turns that off.
    if os.environ.get("RKMH_RAW", "1") != "0" and not os.environ.get("RKMH_CLI_WHOLE_PARSE"):
        out.flush()
        if _device_ingest(ctx, rank, local, world, reads, R["names"], sketch, min_occ, min_matches, min_diff, filter_mode, result_fd, compact_ok):
            ctx.close()
            try:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.destroy_process_group()
            except ImportError:
                pass
            return 0
    # Every rank reads ITS part of the reads only: byte range rank/world of every uncompressed FASTQ file, cut at record starts
    # (rk_reader_open_range); the ranks' record counts put the blocks back in input order.  Text that is not four lines per record
    # (or gzip / FASTA input) cannot be split by bytes: if ANY rank finds that, all of them parse everything and shard by record index.
    parts, ok = [], world > 1 and not os.environ.get("RKMH_CLI_WHOLE_PARSE")
    for path in (reads if ok else []):
        try:
            size = os.path.getsize(path)
            part = api.parse_file_range(path, size * rank // world, size * (rank + 1) // world)
        except (OSError, api.RkmhError):
            ok = False
            break
        if not part["strict"]:
            ok = False
            break
        parts.append(part)
    sharded_ingest = world > 1 and rdist.all_true(ok and len(parts) == len(reads))
    if sharded_ingest:
        Q = parts[0]
        for p in parts[1:]:   # this rank's share of every file, file after file
            nb = int(Q["offsets"][-1])
            Q = {"bases": np.concatenate([Q["bases"][:nb], p["bases"]]), "offsets": np.concatenate([Q["offsets"], p["offsets"][1:] + np.uint64(nb)]),
                 "names": Q["names"] + p["names"], "nseq": Q["nseq"] + p["nseq"],
                 "quals": None if Q["quals"] is None or p["quals"] is None else Q["quals"] + p["quals"]}
        file_counts = [p["nseq"] for p in parts]
        bases, offs = Q["bases"], Q["offsets"]
        lo = 0
    else:
        Q = api.parse_files(reads)  # every rank parses; it classifies only its block of records
        lo, hi = rdist.shard_bounds(Q["nseq"], rank, world)
        offs = Q["offsets"][lo: hi + 1]
        b0, b1 = int(offs[0]), int(offs[-1])
        bases = np.concatenate([Q["bases"][b0:b1], np.zeros(16, np.uint8)])
        offs = offs - np.uint64(b0)
        file_counts = None
    nmine = len(offs) - 1
    counter = None
    if min_occ is not None:
        import torch
        slots = 10000000 if filter_mode else 200000000  # rkmh.cpp:1187 / :739
        dev = "cuda:%d" % local
        # compact depth map (only the slots of index keys) when the output needs min_num up to bound 0 and no read of any rank has
        # more hashes than the sketch keeps (bottom-s selection would need the depth of every hash)
        lens = (offs[1:] - offs[:-1]).astype(np.int64) if len(offs) > 1 else np.zeros(0, dtype=np.int64)
        fits = bool(len(lens) == 0 or (int(lens.max()) <= 1500 and int(sum(max(0, int(lens.max()) - k + 1) for k in ks)) <= sketch))
        compact = rdist.all_true(compact_ok and fits)
        t = torch.zeros(api.Counter.compact_entries(ctx, slots) if compact else slots, dtype=torch.int32, device=dev)
        if t.is_cuda:
            torch.cuda.synchronize()                 # (the fill runs on torch's stream, the count pass on the context's)
        counter = api.Counter(ctx, slots=slots, device_ptr=t.data_ptr(), compact=compact)
        ctx.count_batch(bases, offs, counter)       # pass 1 on this rank's reads
        ctx.synchronize()
        rdist.allreduce_counter(t)                   # RCCL sum over ranks
        if t.is_cuda:
            torch.cuda.synchronize()
        ctx.set_depth_filter(counter, min_occ)
    rows = ctx.classify(bases, offs)
    # every rank formats the lines of ITS reads; rank 0 gathers the text in rank order (file by file when the ingest was sharded by bytes)
    qoff, qbases, quals, qnames = Q["offsets"], Q["bases"], Q.get("quals"), Q["names"]

    def fmt(i_local):
        i = lo + i_local
        r = rows[i_local]
        if not filter_mode:
            return api.format_stream_line(R["names"][int(r[0])], qnames[i], int(r[1]), int(r[2]), int(r[3]), sketch, min_matches, min_diff)
        # classify_and_count_diff_filter scans from max_shared = prev_best = 0 (the stream scan starts at -1)
        if int(r[1]) <= 0:
            shared, diff_ok = 0, 0 > min_diff
        else:
            shared, diff_ok = int(r[1]), (int(r[2]) - (1 if int(r[0]) == 0 else 0)) > min_diff
        # rkmh.cpp:1292-1298; read_min_lens <= 0 implies shared == 0, so the conjunction is the same predicate on exact rows and stays
        # right on rows whose min_num was clamped to 0 (bound 0: only with -D >= 0)
        if (int(r[3]) <= 0 and shared <= 0) or shared < min_matches or not diff_ok:
            return b""
        a, b = int(qoff[i]), int(qoff[i + 1])
        return b">" + qnames[i] + b"\n" + _upper(qbases[a:b]) + b"\n+\n" + (quals[i] if quals is not None else b"") + b"\n"   # rkmh.cpp:1299-1302

    if file_counts is None:
        text = rdist.gather_bytes(b"".join(fmt(i) for i in range(nmine)), dst=0)
        if rank == 0:
            out.write(text)
    else:
        first = 0
        for cnt_f in file_counts:
            text = rdist.gather_bytes(b"".join(fmt(i) for i in range(first, first + cnt_f)), dst=0)
            if rank == 0:
                out.write(text)
            first += cnt_f
    if rank == 0:
        out.flush()
    if counter is not None:
        ctx.set_depth_filter(None, 0)
        counter.destroy()
    ctx.close()
    try:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    except ImportError:
        pass
    return 0


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) <= 1 or argv[1] not in ("stream", "classify", "filter"):
        sys.stderr.write("Usage: python -m rkmh_amd.cli stream|classify|filter [options]   (call, hash, filter -i: use bin/rkmh)\n")
        return 1
    if argv[1] == "filter":
        return main_stream(argv, filter_mode=True)
    if argv[1] == "classify":
        sys.stderr.write("CLASSIFY COMMAND IS TEMPORARILY UNAVAILABLE: TRY rkmh stream INSTEAD.\n")
    return main_stream(argv)


if __name__ == "__main__":
    sys.exit(main())
