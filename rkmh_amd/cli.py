"""`python -m rkmh_amd.cli stream|classify ...` -- the multi-GPU form of the rkmh stream/classify command.

Same flags and stdout as main_stream (/root/reference/src/rkmh.cpp:584-989; option table :626-650; line
format :892).  Launched plainly it uses one GPU; launched under torch.distributed.run it is one process per
GPU: rank 0 sketches the references and broadcasts the sketches over RCCL, every rank classifies a contiguous
block of the reads, the -M path all-reduces the k-mer counter between its two passes
(rkmh.cpp:904-948), and rank 0 prints the lines in input order.  (The single-GPU C++ binary is bin/rkmh.)
"""
import getopt
import sys

import numpy as np

from . import api, dist as rdist

HELP = """rkmh stream|classify -r <refs.fa> -f <reads.fq> [-k <k>]... [-s <sketch>] [-M n] [-I n] [-N n] [-D n]
"""


def main_stream(argv):
    if len(argv) <= 2:
        sys.stderr.write(HELP)
        return 1
    longopts = ["help", "kmer=", "fasta=", "reference=", "sketch-size=", "ref-sketch=", "threads=", "min-kmer-occurence=",
                "min-matches=", "min-diff=", "max-samples=", "pre-reads=", "pre-references=", "read-kmer-map-file=",
                "ref-kmer-map-file=", "in-stream", "output-reads", "merge-sketch"]
    try:
        opts, _ = getopt.getopt(argv[2:], "zmhdk:f:r:s:S:t:M:N:I:R:F:p:q:iD:", longopts)
    except getopt.GetoptError:
        sys.stderr.write(HELP)
        return 1
    refs, reads, ks = [], [], []
    sketch, min_occ, min_matches, min_diff, max_samples = 1000, None, -1, 0, None
    for o, a in opts:
        if o in ("-r", "--reference"): refs.append(a)
        elif o in ("-f", "--fasta"): reads.append(a)
        elif o in ("-k", "--kmer"): ks.append(int(a))
        elif o in ("-s", "--sketch-size"): sketch = int(a)
        elif o in ("-M", "--min-kmer-occurence"): min_occ = int(a)
        elif o in ("-I", "--max-samples"): max_samples = int(a)
        elif o in ("-N", "--min-matches"): min_matches = int(a)
        elif o in ("-D", "--min-diff"): min_diff = int(a)
        elif o in ("-h", "--help", "-d"):
            sys.stderr.write(HELP)
            return 1
    if not ks:
        sys.stderr.write("No kmer size(s) provided. Will use a default kmer size of 16.\n")
        ks = [16]
    if not refs:
        sys.stderr.write("rkmh: at least one -r reference file is required\n")
        return 1
    rank, local, world = rdist.init()
    ctx = api.Context(local)
    R = api.parse_files(refs)
    if R["nseq"] < 1:
        sys.stderr.write("rkmh: no reference sequences found\n")
        return 1
    if rank == 0:
        ctx.set_references(R["bases"], R["offsets"], ks, sketch, max_samples=max_samples)
        sk, ln = ctx.get_reference_sketches()
    else:
        sk = ln = None
    sk, ln = rdist.broadcast_sketches(sk, ln, R["nseq"], sketch, src=0)
    if rank != 0:
        ctx.set_reference_sketches(sk, ln, ks, sketch)
    Q = api.parse_files(reads)  # every rank parses; it classifies only its block
    lo, hi = rdist.shard_bounds(Q["nseq"], rank, world)
    offs = Q["offsets"][lo: hi + 1]
    b0, b1 = int(offs[0]), int(offs[-1])
    bases = np.concatenate([Q["bases"][b0:b1], np.zeros(16, np.uint8)])
    offs = offs - np.uint64(b0)
    counter = None
    if min_occ is not None:
        import torch
        slots = 200000000  # rkmh.cpp:739
        dev = "cuda:%d" % local
        t = torch.zeros(slots, dtype=torch.int32, device=dev)
        counter = api.Counter(ctx, slots=slots, device_ptr=t.data_ptr())
        ctx.count_batch(bases, offs, counter)       # pass 1 on this rank's reads
        ctx.synchronize()
        rdist.allreduce_counter(t)                   # RCCL sum over ranks
        if t.is_cuda:
            torch.cuda.synchronize()
        ctx.set_depth_filter(counter, min_occ)
    rows = ctx.classify(bases, offs)
    allrows = rdist.gather_rows(rows, dst=0)
    if rank == 0:
        out = sys.stdout.buffer
        for i in range(Q["nseq"]):
            r = allrows[i]
            out.write(api.format_stream_line(R["names"][int(r[0])], Q["names"][i], int(r[1]), int(r[2]), int(r[3]), sketch,
                                             min_matches, min_diff))
        out.flush()
    if counter is not None:
        ctx.set_depth_filter(None, 0)
        counter.destroy()
    ctx.close()
    return 0


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) <= 1 or argv[1] not in ("stream", "classify"):
        sys.stderr.write("Usage: python -m rkmh_amd.cli stream|classify [options]   (hash: use bin/rkmh)\n")
        return 1
    if argv[1] == "classify":
        sys.stderr.write("CLASSIFY COMMAND IS TEMPORARILY UNAVAILABLE: TRY rkmh stream INSTEAD.\n")
    return main_stream(argv)


if __name__ == "__main__":
    sys.exit(main())
