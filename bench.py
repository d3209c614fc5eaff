#!/usr/bin/env python3
"""bench.py -- reads classified/sec of the rkmh classify/stream hot path on N MI355X (BASELINE.json metric).

Workload (config C2 of SURVEY.md section 8d, per GPU): 1,000,000 synthetic 150 bp reads drawn from the 182
references of data/all_pave_ref.fa (1 % substitutions, strand flips, 1/1000 reads with an N), k=16, s=1000.
A "step" = one pass of the hot path (rk_classify_batch_device: upper-case, hash, sketch, intersect against
every reference, argmax/diff) over the rank's resident batch.  Reads are sharded over ranks (weak scaling:
every rank owns --reads reads of the global set); reference sketches are built on rank 0 and broadcast over
RCCL once before the timed region.  Inputs are resident in HBM when the timed region starts.

The timed steps rotate over --batches (default 4) DISTINCT resident batches of --reads reads each (4 x 170 MB = 680 MB, more than
the 256 MiB Infinity Cache), so no step finds its input cached from the step before: the bases really stream from HBM.

One JSON line on rank 0 with the contract fields plus `roofline` (HBM bound; algorithmic bytes = 170 B per
read: 150 B bases + 4 B offset + 16 B result) and, at N=1, `cpu_baseline` (the oracle = literal restatement
of src/rkmh.cpp:845-898 with OpenMP, timed on this box's host cores on a bounded sample of the same reads,
and used to check the GPU rows bit-for-bit), `host_path` (rk_classify_batch from pageable host memory: PCIe inclusive) and `e2e`
(bin/rkmh stream on a generated FASTQ: parser + PCIe + kernel + TSV, wall clock of the whole process).  Neither is `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

B_READ = 170.0          # algorithmic bytes per 150 bp read (ASCII input): 150 + 4 + 16
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md)


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box exposes
    256 logical CPUs but grants a 16-CPU quota; more OpenMP threads than that only add contention)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(-(-int(q) // int(p)))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def e2e_stream(n, L, rb, ro, synth):
    """bin/rkmh stream on a generated FASTQ in /tmp: wall clock of the whole process (start-up, context, reference sketches,
    parser, host pipeline, kernel, TSV formatting and writing).  Returns the `e2e` object of the bench line."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "bin", "rkmh")
    ref = os.path.join(ROOT, "tests", "golden", "data", "all_pave_ref.fa.gz")
    tmp = tempfile.mkdtemp(prefix="rkmh_e2e_")
    fq, tsv = os.path.join(tmp, "reads.fq"), os.path.join(tmp, "out.tsv")
    extra = []
    try:
        with open(fq, "wb") as f:
            for lo in range(0, n, 1000000):          # fixed-width records, built 1 M at a time: "@r%09d\n" seq "\n+\n" qual "\n"
                m = min(1000000, n - lo)
                qb, _ = synth.generate_reads_fast(rb, ro, lo, lo + m, read_len=L, threads=min(32, os.cpu_count() or 1))
                rec = np.empty((m, 11 + L + 3 + L + 1), dtype=np.uint8)
                rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 11 + L] = 10; rec[:, 10] = 10
                idx = np.arange(lo, lo + m, dtype=np.int64)
                for d in range(9):
                    rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
                rec[:, 11:11 + L] = qb[: m * L].reshape(m, L)
                rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
                rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
                f.write(rec.tobytes())
        size = os.path.getsize(fq)
        best = None
        def settle():
            # bin/rkmh returns when its output is complete; the kernel takes the GPU context of its worker process apart behind that
            # (0.2-0.8 s).  A timed run must not start beside the last one's teardown: wait until no process called rkmh is left.
            for _ in range(60):
                alive = False
                for pid in os.listdir("/proc"):
                    if pid.isdigit():
                        try:
                            with open("/proc/%s/comm" % pid) as f_:
                                if f_.read().strip() == "rkmh":
                                    alive = True
                                    break
                        except OSError:
                            pass
                if not alive:
                    return
                time.sleep(0.05)

        def fresh_out():                              # a NEW output file per run, opened before the clock starts: truncating the previous run's
            settle()
            if os.path.exists(tsv):                   # gigabyte of cached pages is the harness's cost, not the pipeline's
                os.remove(tsv)
            return open(tsv, "wb")
        for rep in range(2):                          # the second run finds the FASTQ in the page cache, as a pipeline's input would be
            fo = fresh_out()
            t = time.perf_counter()
            r = subprocess.run([exe, "stream", "-r", ref, "-f", fq, "-k", "16", "-s", "1000"], stdout=fo, stderr=subprocess.PIPE,
                               env=dict(os.environ, RKMH_TIMING="1"))
            dt = time.perf_counter() - t
            fo.close()
            if r.returncode != 0:
                return {"error": r.stderr.decode()[-300:]}
            if best is None or dt < best:
                best, stages = dt, [l for l in r.stderr.decode().splitlines() if l.startswith("[rkmh timing]")]
        lines = sum(1 for _ in open(tsv, "rb"))
        res = {"value": n / best, "unit": "reads/s", "reads": n, "fastq_bytes": size, "wall_s": best, "output_lines": lines,
               "stages": [l[len("[rkmh timing] "):].strip() for l in stages],
               "note": "bin/rkmh stream -k 16 -s 1000 on a generated FASTQ, whole process: start-up + reference sketches + parser + PCIe + kernel + TSV"}
        # the same file four times over (-f x 4): what a longer input does to the fixed start-up / tear-down share, and the marginal
        # rate of the pipeline ((4 - 1) n reads in the extra time)
        fo = fresh_out()
        t = time.perf_counter()
        r = subprocess.run([exe, "stream", "-r", ref] + ["-f", fq] * 4 + ["-k", "16", "-s", "1000"], stdout=fo, stderr=subprocess.PIPE)
        dt4 = time.perf_counter() - t
        fo.close()
        if r.returncode == 0 and dt4 > best:
            res["x4"] = {"reads": 4 * n, "wall_s": dt4, "value": 4 * n / dt4, "marginal_reads_per_s": 3 * n / (dt4 - best)}
        # the same two runs with the lines going to /dev/null (one writer thread with fwrite, as for a pipe)
        t1 = []
        for nf in (1, 4):
            settle()
            t = time.perf_counter()
            r = subprocess.run([exe, "stream", "-r", ref] + ["-f", fq] * nf + ["-k", "16", "-s", "1000"], stdout=open(os.devnull, "wb"), stderr=subprocess.PIPE)
            t1.append(time.perf_counter() - t if r.returncode == 0 else None)
        if None not in t1 and t1[1] > t1[0]:
            res["x4_devnull"] = {"wall_s_1": t1[0], "wall_s_4": t1[1], "marginal_reads_per_s": 3 * n / (t1[1] - t1[0]),
                                 "note": "stdout = /dev/null: the ordered single-writer path a pipe gets"}
        # --kmer-cache: the enumeration of the 4^k k-mer universe kept between runs (rk_set_kmer_cache): the "references" stage cold and warm
        kc = os.path.join(tmp, "refs.kmers")
        extra.append(kc)
        ref_s = []
        for _ in range(2):
            r = subprocess.run([exe, "stream", "-r", ref, "-f", fq, "-k", "16", "-s", "1000", "--kmer-cache", kc], stdout=open(os.devnull, "wb"), stderr=subprocess.PIPE,
                               env=dict(os.environ, RKMH_TIMING="1"))
            st_ = [l for l in r.stderr.decode().splitlines() if l.startswith("[rkmh timing] references")]
            ref_s.append(float(st_[0].split()[-2]) if r.returncode == 0 and st_ else None)
        res["kmer_cache"] = {"references_s_enumerated": ref_s[0], "references_s_from_cache": ref_s[1], "file_bytes": os.path.getsize(kc) if os.path.exists(kc) else 0}
        # compressed reads (the reference opens every input with gzopen, rkmh.cpp:238-263): the first reads of the file again with
        # qualities that do not compress to nothing, as plain text, as BGZF (bgzip: independent members, inflated by the front end's
        # workers) and as ordinary single-member gzip (one deflate stream: inflated on the GPU chunk by chunk, rk_gunzip.hip -- or zlib on one thread)
        ng = min(n, 4000000)
        if ng >= 1000:
            import gzip
            import hashlib
            gq, bg, sg = os.path.join(tmp, "gz.fq"), os.path.join(tmp, "gz.bgzf.fq.gz"), os.path.join(tmp, "gz.single.fq.gz")
            extra += [gq, bg, sg]
            rl = 14 + 2 * L + 1
            n1 = min(ng, 1000000)
            with open(fq, "rb") as f, open(gq, "wb") as fo, open(bg, "wb") as fb:
                for lo in range(0, ng, 1000000):
                    m = min(1000000, ng - lo)
                    rec = np.frombuffer(f.read(m * rl), dtype=np.uint8).reshape(m, rl).copy()
                    rec[:, 14 + L:14 + 2 * L] = np.random.default_rng(lo).integers(35, 75, size=(m, L), dtype=np.uint8)
                    raw = rec.tobytes()
                    fo.write(raw)
                    img = synth.bgzf_compress(raw, level=1, threads=min(32, os.cpu_count() or 1))
                    fb.write(img[:-28] if lo + m < ng else img)      # (one end-of-file member, at the end)
                    if lo == 0:
                        with open(sg, "wb") as fs:
                            fs.write(gzip.compress(raw[: n1 * rl], 1))

            def timed(files, env=None):
                # (the quicker of two runs: walls of one process move by 0.1-0.3 s between runs on this pool -- as much as the
                # differences the marginal rates are computed from)
                d_ = None
                for _ in range(2):
                    fo_ = fresh_out()
                    t_ = time.perf_counter()
                    r_ = subprocess.run([exe, "stream", "-r", ref, "-k", "16", "-s", "1000"] + sum((["-f", x] for x in files), []), stdout=fo_, stderr=subprocess.PIPE,
                                        env=dict(os.environ, **(env or {})))
                    dd = time.perf_counter() - t_
                    fo_.close()
                    if r_.returncode != 0:
                        raise RuntimeError(r_.stderr.decode()[-300:])
                    d_ = dd if d_ is None else min(d_, dd)
                h = hashlib.sha256()
                with open(tsv, "rb") as f_:
                    for blk in iter(lambda: f_.read(1 << 24), b""):
                        h.update(blk)
                return d_, h.hexdigest()
            try:
                p1, hp = timed([gq])
                p4, _ = timed([gq] * 4)
                b1, hb = timed([bg], {"RKMH_BGZF_DEVICE": "0"})
                b4, _ = timed([bg] * 4, {"RKMH_BGZF_DEVICE": "0"})
                d1, hd = timed([bg], {"RKMH_BGZF_DEVICE": "1"})
                d4, _ = timed([bg] * 4, {"RKMH_BGZF_DEVICE": "1"})
                s1, hs = timed([sg])
                s4, _ = timed([sg] * 4)
                z1, hz = timed([sg], {"RKMH_GZIP_DEVICE": "0"})
                z4, _ = timed([sg] * 4, {"RKMH_GZIP_DEVICE": "0"})
                res["gz"] = {"reads": ng, "fastq_bytes": os.path.getsize(gq), "bgzf_bytes": os.path.getsize(bg),
                             "plain_wall_s": p1, "bgzf_wall_s": b1, "bgzf_x4_wall_s": b4,
                             "bgzf_marginal_reads_per_s": 3 * ng / (b4 - b1) if b4 > b1 else None,
                             "bgzf_output_identical_to_plain": hb == hp and hd == hp,
                             "bgzf_device_wall_s": d1, "bgzf_device_x4_wall_s": d4,
                             "bgzf_device_marginal_reads_per_s": 3 * ng / (d4 - d1) if d4 > d1 else None,
                             "single_member_reads": n1, "single_member_wall_s": s1,
                             "single_member_marginal_reads_per_s": 3 * n1 / (s4 - s1) if s4 > s1 else None,
                             "single_member_host_wall_s": z1, "single_member_host_marginal_reads_per_s": 3 * n1 / (z4 - z1) if z4 > z1 else None,
                             "single_member_output_identical_on_both_routes": hs == hz,
                             "plain_x4_wall_s": p4, "plain_marginal_reads_per_s": 3 * ng / (p4 - p1) if p4 > p1 else None,
                             "note": "bin/rkmh stream on the same reads as plain FASTQ, as BGZF (level 1, 64 KB members; bgzf_device_*: the default -- "
                                     "the members inflated on the GPU, rk_inflate.hip: a third of a file per job, CRC-32 checked, the text never on "
                                     "the host, names packed on the device; bgzf_*: RKMH_BGZF_DEVICE=0, the workers of the device front end inflate "
                                     "their jobs' members, libdeflate, all but two CPUs) and as ordinary single-member gzip (single_member_*: the default -- "
                                     "the ONE deflate stream inflated on the GPU, rk_gunzip.hip: block headers found by a kernel, a lane per chunk of "
                                     "32 KB, the windows resolved in stream order, CRC-32 and ISIZE checked; single_member_host_*: RKMH_GZIP_DEVICE=0, "
                                     "zlib on its own thread + the block-parallel scanner); marginal = the extra reads of four -f files over one, per "
                                     "extra second of wall clock; every wall is the quicker of two runs"}
                if hs != hz:
                    raise SystemExit("e2e: the device-inflated gzip run printed other bytes than the zlib run")
                if hb != hp or hd != hp:
                    raise SystemExit("e2e: the BGZF run printed other bytes than the plain-text run")
                # the same reads packed once (`rkmh pack`: 2 bits per base + names; qualities dropped) and classified from the packed file
                # (`stream -F`): ~42 bytes per read over the link, nothing parsed, lines formatted from the names in the mapped file
                rkp = os.path.join(tmp, "gz.rkp")
                extra.append(rkp)
                t_ = time.perf_counter()
                rp = subprocess.run([exe, "pack", "-f", gq, "-o", rkp, "--no-quals"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                pack_s = time.perf_counter() - t_
                if rp.returncode != 0:
                    raise RuntimeError(rp.stderr.decode()[-300:])

                def timed_packed(nf, to_file=True):
                    # (the quicker of two runs: one-file walls move by 0.1-0.3 s between runs on this pool, which is the size of the
                    # difference being measured)
                    best_ = None
                    for _ in range(2):
                        fo_ = fresh_out() if to_file else open(os.devnull, "wb")
                        settle()
                        t2 = time.perf_counter()
                        r2 = subprocess.run([exe, "stream", "-r", ref, "-k", "16", "-s", "1000"] + ["-F", rkp] * nf, stdout=fo_, stderr=subprocess.PIPE)
                        d2 = time.perf_counter() - t2
                        fo_.close()
                        if r2.returncode != 0:
                            raise RuntimeError(r2.stderr.decode()[-300:])
                        best_ = d2 if best_ is None else min(best_, d2)
                    h2 = hashlib.sha256()
                    if to_file:
                        with open(tsv, "rb") as f_:
                            for blk in iter(lambda: f_.read(1 << 24), b""):
                                h2.update(blk)
                    return best_, h2.hexdigest()
                k1, hk = timed_packed(1)
                k8, _ = timed_packed(8)
                n1_, _ = timed_packed(1, False)
                n8_, _ = timed_packed(8, False)
                res["packed"] = {"reads": ng, "packed_bytes": os.path.getsize(rkp), "bytes_per_read_in_file": os.path.getsize(rkp) / ng, "pack_s": pack_s,
                                 "wall_s": k1, "x8_wall_s": k8, "marginal_reads_per_s": 7 * ng / (k8 - k1) if k8 > k1 else None,
                                 "devnull_wall_s": n1_, "devnull_x8_wall_s": n8_,
                                 "devnull_marginal_reads_per_s": 7 * ng / (n8_ - n1_) if n8_ > n1_ else None,
                                 "output_identical_to_plain": hk == hp,
                                 "note": "bin/rkmh pack once, then stream -F: per read 37.5 B of 2-bit bases + a 4 B offset go up, 16 B of row come back; "
                                         "marginal = the extra reads of eight -F files over one, per extra second of wall clock (the quicker of two "
                                         "runs each); the lines go to a file on the box's disk (marginal_*: what a user gets -- the file system takes "
                                         "them at 3-5 GB/s) or to /dev/null (devnull_*: what the input side can do)"}
                if hk != hp:
                    raise SystemExit("e2e: the packed run printed other bytes than the plain-text run")
            except RuntimeError as e:
                res["gz"] = {"error": str(e)}
        return res
    finally:
        for x in [fq, tsv] + extra:
            try:
                os.remove(x)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass


def oracle_sketch_long(oracle, arr, k, S, threads):
    """Bottom-S sketch (non-zero hashes, no dedup: rkmh.cpp:822) of ONE long upper-case sequence by the CPU oracle: calc_hashes on
    4 M-window pieces in `threads` threads (ctypes releases the GIL), the S smallest of each piece, then of all.  Checker only."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(arr)
    step = 4 << 20

    def piece(lo):
        hi = min(n, lo + step + k)          # len - k windows per piece (the default window rule): windows starting in [lo, lo + step)
        h = oracle.calc_hashes(arr[lo:hi].tobytes(), [k])
        h = h[h != 0]
        return np.partition(h, S - 1)[:S] if len(h) > S else h
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(piece, range(0, max(n - k, 1), step)))
    h = np.sort(np.concatenate(parts))
    return h[:S]


def c4_full_size(api, synth, genome_mb=3100, nreads=10000000, L=150, check=True):
    """BASELINE config 4 at its own size, whole process: `bin/rkmh filter -k 20 -s 2000` (and with -M 2) of 10 M reads (90 % drawn from
    the genome, 10 % from data/hpv_16_allFasta.fa) against a synthetic genome of 24 sequences with hg38's chromosome lengths (3.1 Gb,
    bases uniform at random) written as FASTA in /tmp.  Each setting
    runs twice (the second finds the files in the page cache); the run with the host parsers (RKMH_RAW=0 RKMH_RAW_REFS=0) is timed
    once beside it and its output must be byte-identical."""
    import hashlib
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "bin", "rkmh")
    tmp = tempfile.mkdtemp(prefix="rkmh_c4_")
    fa, fq = os.path.join(tmp, "genome.fa"), os.path.join(tmp, "reads.fq")
    nthreads = min(32, os.cpu_count() or 1)
    try:
        t0 = time.perf_counter()
        rng = np.random.default_rng(3)
        # 24 sequences with the lengths of hg38's primary chromosomes, scaled to genome_mb million bases in all
        hg38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
                114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
        scale = genome_mb * 1e6 / sum(hg38)
        lut = np.frombuffer(b"ACGT" * 64, dtype=np.uint8)
        parts, goffs = [], [0]
        with open(fa, "wb") as f:
            for c, full in enumerate(hg38):
                chrom = max(1000, int(full * scale))
                s_ = lut[np.frombuffer(rng.bytes(chrom), dtype=np.uint8)]
                f.write(b">chr%s synthetic\n" % (b"X" if c == 22 else b"Y" if c == 23 else b"%d" % (c + 1)))
                for lo in range(0, chrom, 1 << 28):
                    f.write(s_[lo: lo + (1 << 28)].tobytes())
                f.write(b"\n")
                parts.append(s_)
                goffs.append(goffs[-1] + chrom)
        gb = np.concatenate(parts + [np.zeros(16, np.uint8)])
        del parts
        go = np.array(goffs, dtype=np.uint64)
        hpv = api.parse_files([os.path.join(ROOT, "tests", "golden", "data", "hpv_16_allFasta.fa.gz")])
        nh = nreads // 10
        with open(fq, "wb") as f:
            for src_b, src_o, m, tag in ((gb, go, nreads - nh, ord("g")), (hpv["bases"], hpv["offsets"], nh, ord("v"))):
                for lo in range(0, m, 1000000):
                    k = min(1000000, m - lo)
                    qb, _ = synth.generate_reads_fast(src_b, src_o, lo, lo + k, read_len=L, threads=nthreads)
                    rec = np.empty((k, 11 + L + 3 + L + 1), dtype=np.uint8)
                    rec[:, 0] = ord("@"); rec[:, 1] = tag; rec[:, 10] = 10; rec[:, 11 + L] = 10
                    idx = np.arange(lo, lo + k, dtype=np.int64)
                    for d in range(8):
                        rec[:, 9 - d] = 48 + (idx // 10 ** d) % 10
                    rec[:, 11:11 + L] = qb[: k * L].reshape(k, L)
                    rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
                    rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
                    f.write(rec.tobytes())
        gen_s = time.perf_counter() - t0
        checked = None
        if check:
            # the oracle at C4's OWN size: the device's sketches of the largest and the smallest chromosome (248 M and 47 M hashes
            # through the multi-block radix select) against the CPU oracle's, and 55 000 sampled reads -- rows bit-exact, and the
            # filter decisions against what bin/rkmh prints for them (below)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle  # checker only
            import rkmh_amd
            from rkmh_amd import cli
            thr = min(oracle.max_threads(), usable_cpus())
            tc = time.perf_counter()
            cx = rkmh_amd.Context(0)
            try:
                names = cli._references_on_device(cx, [fa], [20], 2000, None, 10000000, True)   # as bin/rkmh filter sets them
                if names is None:
                    raise RuntimeError("the references did not go through the device front end")
                dsk, dln = cx.get_reference_sketches()
                lens = [int(go[i + 1] - go[i]) for i in range(24)]
                which = sorted({int(np.argmax(lens)), int(np.argmin(lens))})
                for c in which:
                    osk = oracle_sketch_long(oracle, gb[int(go[c]): int(go[c + 1])], 20, 2000, thr)
                    if not (int(dln[c]) == len(osk) and (dsk[c, : len(osk)] == osk).all()):
                        raise SystemExit("c4_full_size: ORACLE CHECK FAILED: the device's sketch of sequence %d (%d bases) differs" % (c, lens[c]))
                ng, nv = min(50000, nreads - nh), min(5000, nh)
                sb, so = synth.generate_reads_fast(gb, go, 0, ng, read_len=L, threads=nthreads)
                vb, vo = synth.generate_reads_fast(hpv["bases"], hpv["offsets"], 0, nv, read_len=L, threads=nthreads)
                qb = np.concatenate([sb[: ng * L], vb[: nv * L], np.zeros(16, np.uint8)])
                qo = np.arange(0, (ng + nv) * L + 1, L, dtype=np.uint64)
                want = oracle.classify_stream(qb, qo, [20], 2000, dsk, dln, threads=thr)
                got = cx.classify(qb, qo)
                if not (got == want).all():
                    raise SystemExit("c4_full_size: ORACLE CHECK FAILED: rows of the sampled reads differ")
                expect = set()
                for i in range(ng + nv):
                    if oracle.filter_decision(want[i])[3]:
                        expect.add(b"%c%08d" % (ord("g") if i < ng else ord("v"), i if i < ng else i - ng))
                checked = {"ref_sketches": len(which), "ref_sketch_hashes": [lens[c] - 20 for c in which], "reads": ng + nv, "passing": len(expect),
                           "ranges": (ng, nv), "expect": expect, "seconds": 0.0}
            finally:
                cx.close()
            checked["seconds"] = time.perf_counter() - tc
        del gb
        res = {"genome_bases": int(goffs[-1]), "genome_scale_of_hg38": scale, "fasta_bytes": os.path.getsize(fa), "reads": nreads, "fastq_bytes": os.path.getsize(fq),
               "k": 20, "sketch_size": 2000, "inputs_generated_s": gen_s,
               "note": "bin/rkmh filter, whole process (start-up, 3.1 GB of reference FASTA, sketches, 10 M reads, output); wall_s = device front "
                       "ends (default), host_parser_wall_s = RKMH_RAW=0 RKMH_RAW_REFS=0; outputs compared byte for byte"}

        def run(extra, env, out):
            if os.path.exists(out):
                os.remove(out)
            fo = open(out, "wb")
            t = time.perf_counter()
            r = subprocess.run([exe, "filter", "-r", fa, "-f", fq, "-k", "20", "-s", "2000"] + extra, stdout=fo, stderr=subprocess.PIPE,
                               env=dict(os.environ, RKMH_TIMING="1", **env))
            dt = time.perf_counter() - t
            fo.close()
            if r.returncode != 0:
                raise RuntimeError(r.stderr.decode()[-300:])
            h = hashlib.sha256()
            kept = 0
            with open(out, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
                    kept += blk.count(b">")
            return dt, h.hexdigest(), kept, [l[len("[rkmh timing] "):].strip() for l in r.stderr.decode().splitlines() if l.startswith("[rkmh timing]")]

        def printed_names(out, ng, nv):
            """names of the sampled id ranges among the records bin/rkmh printed"""
            got = set()
            with open(out, "rb") as f:
                for line in f:
                    if line[:1] == b">" and len(line) == 11:
                        nm = line[1:10]
                        if int(nm[1:]) < (ng if nm[:1] == b"g" else nv):
                            got.add(nm)
            return got

        for key, extra in (("plain", []), ("M2", ["-M", "2"])):
            out = os.path.join(tmp, "filter.out")
            best = None
            for _ in range(2):
                dt, dig, kept, stages = run(extra, {}, out)
                if best is None or dt < best[0]:
                    best = (dt, dig, kept, stages)
            if checked is not None: # the decisions bin/rkmh printed for the sampled reads (the last run's file) against the oracle's
                if key == "M2":
                    pass   # (-M at this size has no oracle run: 10 M reads through the two-pass CPU loop; tests cover it in miniature)
                elif printed_names(out, *checked["ranges"]) != checked["expect"]:
                    raise SystemExit("c4_full_size: ORACLE CHECK FAILED: bin/rkmh filter printed other reads of the sample than the oracle passes")
            hdt, hdig, _, _ = run(extra, {"RKMH_RAW": "0", "RKMH_RAW_REFS": "0"}, out)
            res[key] = {"wall_s": best[0], "reads_per_s": nreads / best[0], "reads_passing": best[2], "host_parser_wall_s": hdt,
                        "identical_to_host_parsed_run": best[1] == hdig, "stages": best[3]}
            if best[1] != hdig:
                raise SystemExit("c4_full_size: the device front ends and the host parsers printed different bytes (%s)" % key)
        if checked is not None:
            res["oracle_checked_ref_sketches"] = checked["ref_sketches"]
            res["oracle_checked_ref_sketch_hashes"] = checked["ref_sketch_hashes"]
            res["oracle_checked_reads"] = checked["reads"]
            res["oracle_checked_reads_passing"] = checked["passing"]
            res["oracle_check_s"] = checked["seconds"]
        return res
    except (OSError, MemoryError, RuntimeError) as e:
        return {"error": str(e)[-300:]}
    finally:
        for x in os.listdir(tmp):
            try:
                os.remove(os.path.join(tmp, x))
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass


def config_legs(rkmh_amd, api, synth, dev, n, L, check):
    """Informational legs for the other BASELINE configs (never `value`): c3_panel = config 3's ~270-reference panel (every bundled
    FASTA) on one GPU's resident batch; c4_filter = filter's k = 20, s = 2000 shape, plain and with -M 2; c5_call = rkmh call at
    1000x coverage of HPV16.  Each leg samples rows against the CPU oracle when `check`."""
    import subprocess
    import tempfile
    data = os.path.join(ROOT, "tests", "golden", "data")
    legs = {}
    if check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # checker only

    def kernel_ms(ctx, d_b, d_o, d_out, stream, reps=20, warm=5):
        f = lambda: ctx.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), max_read_len=L, stream=stream)  # noqa: E731
        for _ in range(warm):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def sample_check(out, qb, qo, ks, S, sk, ln):
        if not check:
            return 0
        w = min(n, 4096)
        tot = 0
        for lo_ in (0, n // 2, max(0, n - w)):
            want = oracle.classify_stream(qb, qo[lo_: lo_ + w + 1], ks, S, sk, ln, threads=oracle.max_threads())
            if not (want == out[lo_: lo_ + w]).all():
                raise SystemExit("ORACLE CHECK FAILED in a config leg (k=%s S=%d, reads %d..)" % (ks, S, lo_))
            tot += w
        return tot

    def oracle_sketches(ctx, bases, offs, ks, S):
        """the device's reference sketches, which must equal the oracle's own when the leg is checked"""
        sk, ln = ctx.get_reference_sketches()
        if check:
            osk, oln = oracle.sketch_refs(bases, offs, ks, S, threads=oracle.max_threads())
            if not ((osk == sk).all() and (oln == ln).all()):
                raise SystemExit("ORACLE CHECK FAILED in a config leg: reference sketches (k=%s S=%d)" % (ks, S))
            return osk, oln
        return sk, ln

    stream = torch.cuda.current_stream().cuda_stream
    # ---- config 3's panel: every bundled reference
    files = ["all_pave_ref.fa.gz", "zika.refs.fa.gz", "dengue.fa.gz", "new_refs.fa.gz", "hpv_16.fa.gz", "zika.fa.gz", "yellow_fever.fa.gz",
             "hpv_16_allFasta.fa.gz"]
    panel = api.parse_files([os.path.join(data, f) for f in files])
    pb, po, PR = panel["bases"], panel["offsets"], panel["nseq"]
    ctx = rkmh_amd.Context(dev.index)
    try:
        ctx.set_references(pb, po, [16], 1000)
        sk, ln = oracle_sketches(ctx, pb, po, [16], 1000)
        qb, qo = synth.generate_reads_fast(pb, po, 0, n, read_len=L, threads=min(32, os.cpu_count() or 1))
        d_b = torch.from_numpy(qb).to(dev)
        d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
        d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        ms = kernel_ms(ctx, d_b, d_o, d_out, stream)
        out = d_out.cpu().numpy()
        legs["c3_panel"] = {"references": PR, "k": 16, "sketch_size": 1000, "reads": n, "kernel_ms": ms, "reads_per_s": n / ms * 1e3,
                            "kernel_form": "k-mer-space" if ctx.kmer_form()[0] else "hash-space",
                            "rerouted_rows": int((out[:, 0] < 0).sum()), "oracle_checked_reads": sample_check(out, qb, qo, [16], 1000, sk, ln),
                            "oracle_checked_ref_sketches": int(PR) if check else 0,
                            "note": "BASELINE config 3's panel (every bundled FASTA), one GPU's resident batch of synthetic reads drawn from it"}
    finally:
        ctx.close()
    # ---- config 4's shape: k = 20, s = 2000 against the PaVE panel, plain and with -M 2 (filter's 10 M-slot table)
    pave = api.parse_files([os.path.join(data, "all_pave_ref.fa.gz")])
    rb, ro = pave["bases"], pave["offsets"]
    ctx = rkmh_amd.Context(dev.index)
    try:
        ctx.set_references(rb, ro, [20], 2000)
        sk, ln = oracle_sketches(ctx, rb, ro, [20], 2000)
        qb, qo = synth.generate_reads_fast(rb, ro, 0, n, read_len=L, threads=min(32, os.cpu_count() or 1))
        d_b = torch.from_numpy(qb).to(dev)
        d_o = torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev)
        d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        ms = kernel_ms(ctx, d_b, d_o, d_out, stream)
        out = d_out.cpu().numpy()
        nchk = sample_check(out[:], qb, qo, [20], 2000, sk, ln) if not (out[:, 0] < 0).any() else 0
        slots = 10000000

        def count_ms_of(cnt, reps=5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=stream)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                ctx.count_device(d_b.data_ptr(), d_o.data_ptr(), n, cnt, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        cnt = api.Counter(ctx, slots)
        count_full_ms = count_ms_of(cnt)
        ctx.set_depth_filter(cnt, 2)
        masked_exact_ms = kernel_ms(ctx, d_b, d_o, d_out, stream, reps=10, warm=3)
        ctx.set_depth_filter(None, 0)
        cnt.destroy()
        # as `filter -M 2` runs it (-D >= 0): min_num bound 0, compact depth map, the mask per index key
        ctx.set_min_num_bound(0)
        cnt = api.Counter(ctx, slots, compact=True)
        count_ms = count_ms_of(cnt)
        ctx.set_depth_filter(cnt, 2)
        masked_ms = kernel_ms(ctx, d_b, d_o, d_out, stream, reps=10, warm=3)
        ctx.set_depth_filter(None, 0)
        cnt.destroy()
        ctx.set_min_num_bound(-1)
        # the same batch on the k-mer-space kernel (wide k-mers, k = 17 .. 20): needs the 4^20 k-mer universe enumerated once per reference
        # set (seconds: done unasked only up to k = 18, RKMH_KMER_ENUM_MAXK; a --kmer-cache file keeps it between runs)
        wide = {}
        try:
            os.environ["RKMH_KMER_ENUM_MAXK"] = "20"
            cw = rkmh_amd.Context(dev.index)
            t_en = time.perf_counter()
            cw.set_references(rb, ro, [20], 2000)
            t_en = time.perf_counter() - t_en
            if cw.kmer_form()[0]:
                d_out2 = torch.zeros((n, 4), dtype=torch.int32, device=dev)
                wms = kernel_ms(cw, d_b, d_o, d_out2, stream)
                if not (d_out2.cpu().numpy() == out).all():      # (`out`: the hash-space kernel's rows of the plain run above)
                    raise SystemExit("c4_filter: the k-mer-space kernel (wide k-mers) and the hash-space kernel disagree")
                wide = {"kmer_space_kernel_ms": wms, "kmer_space_reads_per_s": n / wms * 1e3, "kmer_space_set_references_s": t_en}
            cw.close()
        finally:
            os.environ.pop("RKMH_KMER_ENUM_MAXK", None)
        legs["c4_filter"] = {"references": pave["nseq"], "k": 20, "sketch_size": 2000, "reads": n, "kernel_ms": ms, "reads_per_s": n / ms * 1e3, **wide,
                             "M2_slots": slots, "M2_count_pass_ms": count_ms, "M2_masked_classify_ms": masked_ms, "M2_min_num_bound": 0,
                             "M2_count_pass_full_table_ms": count_full_ms, "M2_masked_classify_exact_min_num_ms": masked_exact_ms,
                             "oracle_checked_ref_sketches": int(pave["nseq"]) if check else 0,
                             "rerouted_rows": int((out[:, 0] < 0).sum()), "oracle_checked_reads": nchk,
                             "note": "BASELINE config 4's kernel shape (filter: k = 20, s = 2000; hash-space kernel) on one resident batch against the "
                                     "PaVE panel; full_size = the whole command at the config's own size (3.1 Gb genome, 10 M reads)"}
    finally:
        ctx.close()
    # ---- config 5: rkmh call, whole process, 1000x coverage of HPV16 with planted variants
    exe = os.path.join(ROOT, "bin", "rkmh")
    tmp = tempfile.mkdtemp(prefix="rkmh_c5_")
    try:
        h16 = api.parse_files([os.path.join(data, "hpv_16.fa.gz")])
        ref = bytes(h16["bases"][: int(h16["offsets"][1])]).upper()
        mut = bytearray(ref)
        for pos, alt in ((500, b"A"), (1200, b"C"), (2503, b"G"), (4000, b"T"), (6100, b"A")):
            mut[pos] = alt[0] if mut[pos] != alt[0] else b"ACGT"[(b"ACGT".index(alt) + 1) % 4]
        for pos in (7000, 3100):
            del mut[pos]
        rng = np.random.default_rng(5)
        nr = 1000 * len(ref) // 150
        st = rng.integers(0, len(mut) - 150, size=nr)
        arr = np.frombuffer(bytes(mut), dtype=np.uint8)
        reads = arr[st[:, None] + np.arange(150)[None, :]].copy()
        noise = rng.random(reads.shape) < 0.005
        reads[noise] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(noise.sum()))]
        fa, fq = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
        with open(fa, "wb") as f:
            f.write(b">HPV16\n" + ref + b"\n")
        with open(fq, "wb") as f:
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, reads[i].tobytes(), b"I" * 150) for i in range(nr)))
        best, rows = None, 0
        for _ in range(2):
            t = time.perf_counter()
            r = subprocess.run([exe, "call", "-r", fa, "-f", fq, "-k", "12"], capture_output=True)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                legs["c5_call"] = {"error": r.stderr.decode()[-300:]}
                break
            rows = sum(1 for l in r.stdout.decode().splitlines() if l and not l.startswith("#"))
            best = dt if best is None or dt < best else best
        else:
            legs["c5_call"] = {"reads": nr, "k": 12, "wall_s": best, "reads_per_s": nr / best, "vcf_rows": rows,
                               "note": "bin/rkmh call -k 12, whole process, 1000x coverage of HPV16 (5 planted SNPs, 2 planted deletions, 0.5 % noise); "
                                       "rows vs the oracle are a GPU test (test_call_matches_oracle / test_call_at_c5_scale)"}
    finally:
        for x in os.listdir(tmp):
            os.remove(os.path.join(tmp, x))
        os.rmdir(tmp)
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--spinup-seconds", type=float, default=0.3,
                    help="untimed launches before the warm-up steps until the GPU clocks have ramped (a 1 ms kernel measured "
                         "during the first ~50 ms after idle runs 4-12 %% slower); 0 disables")
    ap.add_argument("--reads", type=int, default=1000000, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time (0 disables)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the timed steps rotate over (>= 4 x 170 MB defeats the 256 MiB Infinity Cache)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive rk_classify_batch figure (N=1 only)")
    ap.add_argument("--no-depth-filter", action="store_true", help="skip the -M figures (count pass + masked classify at 200 M slots; N=1 only)")
    ap.add_argument("--e2e-reads", type=int, default=16000000, help="reads of the generated FASTQ for the bin/rkmh stream end-to-end figure (0 disables; N=1 only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the informational legs for BASELINE configs 3, 4 and 5 (N=1 only)")
    ap.add_argument("--no-c4-full", action="store_true", help="skip config 4 at full size (3.1 Gb genome + 10 M reads generated in /tmp, ~40 s)")
    ap.add_argument("--c4-genome-mb", type=int, default=3100, help="config 4 at full size: bases of the synthetic genome, in millions")
    ap.add_argument("--c4-reads", type=int, default=10000000, help="config 4 at full size: reads")
    ap.add_argument("--c3-total-reads", type=int, default=100000000,
                    help="N > 1 runs BASELINE config 3: this many reads IN ALL (strong scaling: each rank classifies total / N) against every bundled reference")
    ap.add_argument("--c3-launch-reads", type=int, default=12500000, help="N > 1: reads per kernel launch (a rank's shard is resident as batches of at most this many)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the N ranks ourselves, as CHILD processes of torch.distributed.run, before anything in this
        # process has touched the GPU (a process that has initialised HIP must never exec or fork workers)
        import socket
        import subprocess
        with socket.socket() as sk_:
            sk_.bind(("127.0.0.1", 0))
            port = sk_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import rkmh_amd
    from rkmh_amd import api, dist as rdist, synth

    rank, local, world = rdist.init()
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    if os.environ.get("RKMH_BENCH_ONE_DEVICE"):   # plumbing test: several ranks on one GPU (with RKMH_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = rkmh_amd.Context(local)

    ks, S, L = [16], 1000, 150
    data = os.path.join(ROOT, "tests", "golden", "data")
    # N = 1: BASELINE config 2 (stream: 1 M-read batches against the PaVE panel), weak scaling never applies.  N > 1: config 3 --
    # --c3-total-reads reads IN ALL against every bundled reference (266 sequences), sharded over the ranks: STRONG scaling.
    c3_mode = world > 1
    ref_files = (["all_pave_ref.fa.gz", "zika.refs.fa.gz", "dengue.fa.gz", "new_refs.fa.gz", "hpv_16.fa.gz", "zika.fa.gz", "yellow_fever.fa.gz",
                  "hpv_16_allFasta.fa.gz"] if c3_mode else ["all_pave_ref.fa.gz"])
    refs = api.parse_files([os.path.join(data, f) for f in ref_files])
    rb, ro, R = refs["bases"], refs["offsets"], refs["nseq"]
    if rank == 0:
        ctx.set_references(rb, ro, ks, S)                 # sketched on this GPU
        sk, ln = ctx.get_reference_sketches()
    else:
        sk = ln = None
    if world > 1:
        torch.cuda.synchronize()
        torch.distributed.barrier()
    t_bc = time.perf_counter()
    sk, ln = rdist.broadcast_sketches(sk, ln, R, S, src=0,                 # RCCL over xGMI, once
                                      device=dev if os.environ.get("RKMH_DIST_BACKEND", "nccl") == "nccl" else "cpu")
    broadcast_ms = (time.perf_counter() - t_bc) * 1e3
    if rank != 0:
        ctx.set_reference_sketches(sk, ln, ks, S)

    if c3_mode:
        total = a.c3_total_reads
        my_lo, my_hi = rdist.shard_bounds(total, rank, world)
        per = max(1, min(a.c3_launch_reads, 28000000 * 150 // L))   # (32-bit base offsets inside one launch)
        bounds = [(x, min(my_hi, x + per)) for x in range(my_lo, my_hi, per)]
        nb = len(bounds)
        n = my_hi - my_lo                                  # reads of this rank per step (one step = its whole shard)
    else:
        n = a.reads
        nb = max(1, a.batches)
        lo = rank * n * nb          # (N = 1: nb consecutive batches of n reads)
        bounds = [(lo + b * n, lo + (b + 1) * n) for b in range(nb)]
    ns = [hi_ - lo_ for lo_, hi_ in bounds]
    d_bs, d_os, d_outs, qbs, qos = [], [], [], [], []
    for b, (lo_, hi_) in enumerate(bounds):
        qb, qo = synth.generate_reads_fast(rb, ro, lo_, hi_, read_len=L, threads=min(32, os.cpu_count() or 1))
        if not c3_mode or b == 0:
            qbs.append(qb); qos.append(qo)                 # (config 3: only the first batch stays on the host, for the oracle samples)
        d_bs.append(torch.from_numpy(qb).to(dev))
        d_os.append(torch.from_numpy(qo.astype(np.int64)).to(torch.int32).to(dev))
        d_outs.append(torch.zeros((hi_ - lo_, 4), dtype=torch.int32, device=dev))
        del qb, qo
    qb, qo = qbs[0], qos[0]
    torch.cuda.synchronize()                         # (uploads and fills above ran on the default stream)
    tstream = torch.cuda.Stream(device=dev)          # the kernels AND the timing events go on this stream
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    step_no = [0]

    def step():
        if c3_mode:     # one step = this rank's whole shard of the 100 M reads
            for b in range(nb):
                ctx.classify_device(d_bs[b].data_ptr(), d_os[b].data_ptr(), ns[b], d_outs[b].data_ptr(), max_read_len=L, stream=stream)
            return
        b = step_no[0] % nb
        step_no[0] += 1
        ctx.classify_device(d_bs[b].data_ptr(), d_os[b].data_ptr(), n, d_outs[b].data_ptr(), max_read_len=L, stream=stream)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if a.spinup_seconds > 0:          # clock ramp: not part of the W warm-up steps, not timed
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < a.spinup_seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64,
                      device=dev if (world == 1 or torch.distributed.get_backend() == "nccl") else "cpu")
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))   # HIP events on the launch stream

    if any(bool((d[:, 0] < 0).any().item()) for d in d_outs):
        raise SystemExit("fused path flagged reads for rerouting: the benchmark batch must be fused-eligible")
    outs = [d.cpu().numpy() for d in (d_outs[:1] if c3_mode else d_outs)]
    out = outs[0]
    c3_checked = 0
    if c3_mode and a.cpu_seconds > 0:
        # every rank samples rows of its first launch against the CPU oracle (sketches from the oracle too); the counts are summed
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle  # checker only
        thr = a.cpu_threads or max(1, min(oracle.max_threads(), usable_cpus()) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))
        osk, oln = oracle.sketch_refs(rb, ro, ks, S, threads=thr)
        ok = bool((osk == sk).all() and (oln == ln).all())
        w = min(ns[0], 4096)
        for lo_ in sorted({0, max(0, ns[0] // 2 - w), max(0, ns[0] - w)}):
            want = oracle.classify_stream(qb, qo[lo_: lo_ + w + 1], ks, S, osk, oln, threads=thr)
            ok = ok and bool((want == out[lo_: lo_ + w]).all())
            c3_checked += w
        if not rdist.all_true(ok):
            raise SystemExit("ORACLE CHECK FAILED on some rank: GPU rows or reference sketches differ from the CPU oracle")
        c3_checked = sum(rdist.all_gather_int(c3_checked))

    if rank == 0:
        value = (a.c3_total_reads if c3_mode else n) * a.steps / elapsed
        achieved = B_READ * n / (kern_ms * 1e-3) / 1e9     # (config 3: kern_ms spans the launches of one step = this rank's n reads)
        # Offline PMC figures (rocprofv3 --pmc passes of this same command, tools/profile.sh -> tools/make_pmc_json.py) are only
        # quoted while they describe THIS build: same kernel sources (stamp), same reads per launch.  Otherwise null.
        traffic = None
        valu = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                from rkmh_amd.stamp import kernel_source_stamp
                j = json.load(open(pmc))
                if j.get("reads_per_launch") == n and j.get("kernel_source_stamp") == kernel_source_stamp():
                    traffic = j.get("hbm_bytes_per_launch")
                    if j.get("valu_insts_per_launch"):
                        # the binding resource: a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles
                        props = api.device_props(local)
                        simds = 4 * props["compute_units"]
                        ghz = props["clock_khz"] / 1e6   # peak engine clock (kHz -> GHz); the kernel holds ~99 % of it
                        valu = {"insts_per_read": j["valu_insts_per_launch"] / n,
                                "issue_frac": j["valu_insts_per_launch"] * 4.0 / (simds * kern_ms * 1e-3 * ghz * 1e9),
                                "source": "offline estimate: SQ_INSTS_VALU of %s x 4 cycles / (%d SIMDs x live kernel_ms x %.2f GHz peak clock)"
                                          % (j.get("source", "profiles/pmc_latest.json").split(" ")[0], simds, ghz)}
            except (OSError, ValueError, KeyError) as e:   # unreadable / malformed file: no offline figures, and say so
                sys.stderr.write("bench.py: ignoring profiles/pmc_latest.json (%s)\n" % e)
                traffic = None
                valu = None
        res = {
            "metric": "reads classified/sec", "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if c3_mode else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("classify (C3): %d synthetic %d bp reads IN ALL vs every bundled viral reference (%d sequences), k=16 s=1000, "
                                    "reads sharded over %d GPUs; one step = one pass over all of them" % (a.c3_total_reads, L, R, world)) if c3_mode else
                                   ("stream (C2): %d synthetic %d bp reads per GPU vs data/all_pave_ref.fa (%d refs), k=16 s=1000" % (n, L, R)),
                       "reads_per_gpu": n, "read_len": L, "k": 16, "sketch_size": S,
                       "references": R, "parallelism": "reads sharded over %d rank(s); ref sketches RCCL-broadcast once" % world,
                       "spinup_seconds": a.spinup_seconds},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_classify_kmer" if ctx.kmer_form()[0] else "k_classify_tile", "kernel_ms": kern_ms,
                         "bytes_per_read": B_READ, "valu": valu,
                         # which form of the fused kernel ran: "k-mer-space" = reads are filtered and matched by packed k-mer after the
                         # whole 4^k k-mer universe was hashed once when the references were set; "hash-space" = every
                         # window is hashed in the kernel
                         "kernel_form": "k-mer-space" if ctx.kmer_form()[0] else "hash-space"},
        }
        if c3_mode:
            res["config"].update({"reads_total": a.c3_total_reads, "launches_per_step": nb, "sketch_broadcast_ms": broadcast_ms})
            if c3_checked:
                res["oracle_checked_reads"] = c3_checked
                res["oracle_checked_ref_sketches"] = int(R)
        if world == 1 and a.cpu_seconds > 0:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle  # CPU baseline + checker only
            thr = a.cpu_threads or min(oracle.max_threads(), usable_cpus())
            # the reference sketches the rows are checked against are the ORACLE's own (sketched here from the reference bases), and the
            # device's must equal them -- the check below must not be able to pass on a wrong sketch
            osk, oln = oracle.sketch_refs(rb, ro, ks, S, threads=thr)
            if not ((osk == sk).all() and (oln == ln).all()):
                raise SystemExit("ORACLE CHECK FAILED: the device's reference sketches differ from the CPU oracle's")
            sk, ln = osk, oln
            res["oracle_checked_ref_sketches"] = int(R)
            probe = min(n, 4 * thr)
            t = time.perf_counter()
            oracle.classify_stream(qb, qo[: probe + 1], ks, S, sk, ln, threads=thr)
            rate = probe / max(time.perf_counter() - t, 1e-6)
            m = int(max(min(n, rate * a.cpu_seconds), min(n, 1000)))
            t = time.perf_counter()
            want = oracle.classify_stream(qb, qo[: m + 1], ks, S, sk, ln, threads=thr)
            dt = time.perf_counter() - t
            if not (want == out[:m]).all():
                raise SystemExit("ORACLE CHECK FAILED: GPU rows differ from the CPU oracle")
            # every OTHER timed batch too: windows of consecutive reads spread over the batch (its first and its last reads included,
            # where tiles are cut differently), so a row that is only wrong in batches 1.. cannot pass
            checked_other = 0
            for b in range(1, nb):
                w = min(n, 8192)
                starts = sorted(set([0, max(0, n - w)] + [int(x) for x in np.random.default_rng(1000 + b).integers(0, max(1, n - w), size=6)]))
                for lo_ in starts:
                    wb = oracle.classify_stream(qbs[b], qos[b][lo_: lo_ + w + 1], ks, S, sk, ln, threads=thr)
                    if not (wb == outs[b][lo_: lo_ + w]).all():
                        raise SystemExit("ORACLE CHECK FAILED: GPU rows of batch %d differ from the CPU oracle (reads %d..%d)" % (b, lo_, lo_ + w))
                    checked_other += w
            t = time.perf_counter()
            m1 = max(min(m, int(m / thr * 4)), 1)
            oracle.classify_stream(qb, qo[: m1 + 1], ks, S, sk, ln, threads=1)
            dt1 = time.perf_counter() - t
            res["cpu_baseline"] = {"value": m / dt, "unit": "reads/s", "cores": thr, "kind": "port", "cpu_model": cpu_model(),
                                   "sample": "first %d reads of the same batch, OpenMP x%d, refs pre-sketched; %.1f s" % (m, thr, dt),
                                   "single_thread_value": m1 / dt1,
                                   "oracle_check": "GPU rows bit-exact vs the CPU oracle on the first %d reads of batch 0 and on %d reads sampled from the "
                                                   "other %d timed batches (oracle-consistent; the mkmh policies are unpinned by any reference "
                                                   "artefact, DESIGN.md section 0)" % (m, checked_other, nb - 1),
                                   "oracle_checked_reads": m + checked_other}
        if world == 1 and not a.no_host_path:
            # PCIe-inclusive: the same batches from PAGEABLE host memory through rk_classify_batch (pinned staging, H2D, kernel,
            # D2H overlapped chunk by chunk).  Never the reported value.
            hb = np.concatenate([x[: n * L] for x in qbs] + [np.zeros(16, np.uint8)])
            ho = np.arange(nb * n + 1, dtype=np.uint64) * np.uint64(L)
            ctx.classify(hb[: n * L + 16], ho[: n + 1])          # warm-up: staging buffers, first-touch
            t = time.perf_counter()
            hout = ctx.classify(hb, ho)
            dt = time.perf_counter() - t
            if not all((hout[b * n:(b + 1) * n] == outs[b]).all() for b in range(nb)):
                raise SystemExit("host path rows differ from the resident path")
            # the same from PAGE-LOCKED buffers (rk_host_alloc: what the FASTQ front end fills): DMA in place, no staging copy
            pb, po = api.pinned_array(hb.shape, np.uint8), api.pinned_array((nb * n, 4), np.int32)
            pb.array[:] = hb
            ctx.classify(pb.array, ho, out=po.array)
            t = time.perf_counter()
            ctx.classify(pb.array, ho, out=po.array)
            dtp = time.perf_counter() - t
            if not (po.array == hout).all():
                raise SystemExit("host path (page-locked) rows differ from the resident path")
            res["host_path"] = {"value": nb * n / dtp, "unit": "reads/s", "gbytes_per_s_h2d": nb * n * (L + 4) / dtp / 1e9, "reads": nb * n,
                                "pageable_value": nb * n / dt, "pageable_gbytes_per_s": nb * n * (L + 4) / dt / 1e9,
                                "note": "rk_classify_batch, PCIe inclusive (H2D + kernel + D2H overlapped chunk by chunk): value = from page-locked "
                                        "host buffers (rk_host_alloc), read by DMA in place; pageable_value = from ordinary memory, which the "
                                        "library page-locks for the call (hipHostRegister) or, failing that, copies through its staging buffers"}
            del pb, po
        if world == 1 and not a.no_depth_filter:
            # -M (rkmh.cpp:904-917) on the first batch, the reference's 200 M-slot table: pass 1 (count every window's hash) and the
            # masked classification that reads it back.  Informational: never the reported value.
            slots = 200000000
            cnt = api.Counter(ctx, slots)

            def timed(f, reps=10, warm=3):
                for _ in range(warm):
                    f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    f()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps
            count_full_ms = timed(lambda: ctx.count_device(d_bs[0].data_ptr(), d_os[0].data_ptr(), n, cnt, stream=stream))
            ctx.set_depth_filter(cnt, 2)
            masked_exact_ms = timed(lambda: ctx.classify_device(d_bs[0].data_ptr(), d_os[0].data_ptr(), n, d_outs[0].data_ptr(), max_read_len=L, stream=stream))
            ctx.set_depth_filter(None, 0)
            cnt.destroy()
            # what `stream -M 2` runs by default (no -N: num_mins only meets `num_mins <= -1`, rkmh.cpp:938): min_num bound 0, the
            # compact depth map (only the slots of index keys), the mask applied per key on the k-mer-space kernel
            ctx.set_min_num_bound(0)
            ccnt = api.Counter(ctx, slots, compact=True)
            count_ms = timed(lambda: ctx.count_device(d_bs[0].data_ptr(), d_os[0].data_ptr(), n, ccnt, stream=stream))
            ctx.set_depth_filter(ccnt, 2)
            masked_ms = timed(lambda: ctx.classify_device(d_bs[0].data_ptr(), d_os[0].data_ptr(), n, d_outs[0].data_ptr(), max_read_len=L, stream=stream))
            ctx.set_depth_filter(None, 0)
            entries = ccnt.entries
            ccnt.destroy()
            checked = 0
            if a.cpu_seconds > 0:
                # both passes on a sub-batch against the oracle's two-pass loop (min_num clamped to the bound), compact and full maps
                import oracle
                cores = a.cpu_threads or min(oracle.max_threads(), usable_cpus())
                m = min(n, 65536)
                sub_o = np.ascontiguousarray(qos[0][: m + 1])
                sub_b = np.ascontiguousarray(qbs[0][: int(sub_o[-1]) + 16])
                want = oracle.classify_stream(sub_b, sub_o, ks, S, sk, ln, threads=cores, min_kmer_occ=2, counter_slots=slots)
                for compact, bound in ((True, 0), (False, 0), (False, 3), (False, -1)):
                    ctx.set_min_num_bound(bound)
                    c2 = api.Counter(ctx, slots, compact=compact)
                    ctx.count_batch(sub_b, sub_o, c2)
                    ctx.set_depth_filter(c2, 2)
                    got = ctx.classify(sub_b, sub_o)
                    ctx.set_depth_filter(None, 0)
                    c2.destroy()
                    w = want.copy()
                    if bound >= 0:
                        w[:, 3] = np.minimum(w[:, 3], bound)
                    if not (got == w).all():
                        raise SystemExit("-M rows differ from the oracle (compact=%s bound=%d)" % (compact, bound))
                    checked += m
            ctx.set_min_num_bound(-1)
            res["depth_filter"] = {"slots": slots, "count_pass_ms": count_ms, "masked_classify_ms": masked_ms, "reads": n,
                                   "min_num_bound": 0, "compact_map_entries": entries,
                                   "count_pass_full_table_ms": count_full_ms, "masked_classify_exact_min_num_ms": masked_exact_ms,
                                   "oracle_checked_reads": checked,
                                   "note": "-M on one resident 1 M-read batch as `stream -M 2` runs it: min_num bound 0 (rk_set_min_num_bound), pass 1 into "
                                           "the compact depth map (only the slots of index keys), the mask per index key on the k-mer-space kernel; "
                                           "*_full_table / *_exact_min_num = the 200 M-slot table and the per-window slot lookup (what -N >= 0 with a "
                                           "large bound, or RKMH_EXACT_MIN_NUM=1, still runs)"}
        if world == 1 and not a.no_configs:
            res.update(config_legs(rkmh_amd, api, synth, dev, n, L, a.cpu_seconds > 0))
            if not a.no_c4_full and "c4_filter" in res:
                res["c4_filter"]["full_size"] = c4_full_size(api, synth, a.c4_genome_mb, a.c4_reads, check=a.cpu_seconds > 0)
        if world == 1 and a.e2e_reads > 0:
            res["e2e"] = e2e_stream(a.e2e_reads, L, rb, ro, synth)
        print(json.dumps(res))
        sys.stdout.flush()
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
