"""Synthetic workloads of SURVEY.md section 8(d) (C2/C3): 150 bp reads drawn from a reference panel.

Read i is a pure function of (seed + i) through a splitmix64 stream, so any shard [lo, hi) of the global
read set can be generated independently on any rank:
  draw 0: reference (uniform), draw 1: start (uniform in [0, len-L]), draw 2: strand (bit 0),
  draw 3: 1-in-1000 reads get one 'N', draw 4: its position, draws 5..5+L-1: per-base substitution
  (1 % rate, uniform over the three other bases; non-ACGT reference bases are copied unchanged).
Names are r%09d, qualities '+' * L (as in the reference's data/z1.fq).
"""
import numpy as np

SEED = 0x726B6D68  # "rkmh"
_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def _draw(state0, j):
    """j-th output (0-based) of the splitmix64 stream whose initial state is state0 (vectorised)."""
    with np.errstate(over="ignore"):
        return _mix(state0 + _G * np.uint64(j + 1))


_UP = np.arange(256, dtype=np.uint8)
_UP[ord("a"): ord("z") + 1] -= 32
_COMP = np.arange(256, dtype=np.uint8)
for a, b in (("A", "T"), ("C", "G")):
    _COMP[ord(a)], _COMP[ord(b)] = ord(b), ord(a)
_CODE = np.full(256, 4, dtype=np.uint8)
for i, ch in enumerate("ACGT"):
    _CODE[ord(ch)] = i
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def generate_reads(ref_bases, ref_offsets, lo, hi, read_len=150, seed=SEED, chunk=1 << 16):
    """Reads [lo, hi) of the global set -> (uint8 bases [(hi-lo)*read_len + pad], uint64 offsets)."""
    ref_offsets = np.asarray(ref_offsets, dtype=np.uint64)
    nref = len(ref_offsets) - 1
    lens = (ref_offsets[1:] - ref_offsets[:-1]).astype(np.int64)
    if (lens < read_len).any():
        raise ValueError("every reference must be at least read_len long")
    up = _UP[np.asarray(ref_bases[: int(ref_offsets[-1])], dtype=np.uint8)]
    n = hi - lo
    out = np.zeros(n * read_len + 16, dtype=np.uint8)
    L = read_len
    with np.errstate(over="ignore"):
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            idx = np.arange(lo + c0, lo + c1, dtype=np.uint64)
            st = np.uint64(seed) + idx
            ref = (_draw(st, 0) % np.uint64(nref)).astype(np.int64)
            span = (lens[ref] - L + 1).astype(np.uint64)
            start = (_draw(st, 1) % span).astype(np.int64) + ref_offsets[ref].astype(np.int64)
            flip = (_draw(st, 2) & np.uint64(1)).astype(bool)
            hasn = (_draw(st, 3) % np.uint64(1000)) == 0
            npos = (_draw(st, 4) % np.uint64(L)).astype(np.int64)
            reads = up[start[:, None] + np.arange(L, dtype=np.int64)[None, :]]
            d = _mix(st[:, None] + _G * np.arange(6, 6 + L, dtype=np.uint64)[None, :])   # draws 5..5+L-1
            rr, cc = np.nonzero((d % np.uint64(100)) == 0)
            code = _CODE[reads[rr, cc]]
            ok = code < 4
            rr, cc, code = rr[ok], cc[ok], code[ok]
            newc = (code.astype(np.uint64) + np.uint64(1) + ((d[rr, cc] >> np.uint64(32)) % np.uint64(3))) % np.uint64(4)
            reads[rr, cc] = _ACGT[newc.astype(np.int64)]
            if flip.any():
                reads[flip] = _COMP[reads[flip][:, ::-1]]
            rows = np.nonzero(hasn)[0]
            reads[rows, npos[rows]] = ord("N")
            out[c0 * L: c1 * L] = reads.reshape(-1)
    offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    return out, offsets


def generate_reads_fast(ref_bases, ref_offsets, lo, hi, read_len=150, seed=SEED, threads=8):
    """Same stream as generate_reads, produced by the C++ twin in librkmh_amd.so (rk_synth_reads)."""
    import ctypes as C
    from . import api
    lib = api.load_library()
    ref_offsets = np.ascontiguousarray(ref_offsets, dtype=np.uint64)
    ref_bases = np.ascontiguousarray(ref_bases, dtype=np.uint8)
    n = hi - lo
    out = np.zeros(n * read_len + 16, dtype=np.uint8)
    rc = lib.rk_synth_reads(ref_bases.ctypes.data_as(C.POINTER(C.c_uint8)), ref_offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                            len(ref_offsets) - 1, lo, hi, read_len, seed, out.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
    if rc != 0:
        raise ValueError("rk_synth_reads failed (%d)" % rc)
    return out, np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)


def read_names(lo, hi):
    return [b"r%09d" % i for i in range(lo, hi)]


def write_fastq(path, bases, offsets, names):
    with open(path, "wb") as f:
        for i, nm in enumerate(names):
            s = bytes(bases[int(offsets[i]): int(offsets[i + 1])])
            f.write(b"@" + nm + b"\n" + s + b"\n+\n" + b"+" * len(s) + b"\n")


def synthetic_panel(nref=182, min_len=7100, max_len=8104, seed=SEED ^ 0xFFFF):
    """A stand-in reference panel (i.i.d. uniform bases) for boxes without the bundled FASTA fixtures."""
    with np.errstate(over="ignore"):
        st = np.uint64(seed) + np.arange(nref, dtype=np.uint64)
        lens = (min_len + (_draw(st, 0) % np.uint64(max_len - min_len + 1))).astype(np.int64)
        offs = np.zeros(nref + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens).astype(np.uint64)
        tot = int(offs[-1])
        z = _mix(np.uint64(seed) * np.uint64(31) + np.arange(tot, dtype=np.uint64) * _G)
    bases = np.zeros(tot + 16, dtype=np.uint8)
    bases[:tot] = _ACGT[(z >> np.uint64(33)) % np.uint64(4)]
    return bases, offs


def bgzf_compress(data, level=1, block=0xff00, threads=8, strategy=0, mem_level=8):
    """`data` (bytes-like) as a BGZF file image: independent gzip members of at most `block` bytes of text each, the 'BC' extra
    field carrying each member's length, the empty end-of-file member last -- what `bgzip` writes (SAM specification, section 4.1).
    Members are deflated in `threads` threads (zlib releases the GIL).  strategy / mem_level: zlib's (Z_FIXED, Z_RLE, Z_HUFFMAN_ONLY ...;
    a small mem_level means many deflate blocks per member) -- for tests of inflaters."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(data)

    def member(lo):
        chunk = mv[lo: lo + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        body = co.compress(chunk) + co.flush()
        if len(body) + 26 > 65536:                       # incompressible: store it
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            body = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        parts = list(ex.map(member, range(0, len(mv), block)))
    parts.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")
    return b"".join(parts)
