"""Packed reads on the device (rk_packed.hip) and through the command line: `rkmh pack` + `stream|filter -F` print, byte for byte, what the
same commands print for the FASTQ text the file was packed from -- reads with N and IUPAC codes, lower case, ragged lengths from 1 base to
beyond the fused kernel's limit, several blocks and files, with and without -M, with and without kept qualities."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(data_dir, n, seed):
    from rkmh_amd import api, synth
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    qb, qo = synth.generate_reads_fast(refs["bases"], refs["offsets"], 0, n, read_len=150, threads=4)
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        s = bytearray(qb[int(qo[i]):int(qo[i + 1])])
        r = rng.random()
        if r < 0.05:
            s[int(rng.integers(0, 150))] = ord("N")
        elif r < 0.08:
            s = s[: int(rng.integers(1, 150))]
        elif r < 0.10:
            s = bytearray(bytes(s).lower())
        elif r < 0.11:
            s[40:44] = b"RYKM"
        elif r < 0.112:
            s = s * 12                                   # 1800 bases: beyond the fused kernel, the general path on the expanded bases
        recs.append((b"p%07d" % i, bytes(s), bytes(rng.integers(35, 75, size=len(s), dtype=np.uint8))))
    return recs


def _run(cmd, env=None):
    r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    return r.stdout


def test_cli_packed_reads_print_what_the_text_prints(root, data_dir, tmp_path):
    recs = _reads(data_dir, 60000, 1)
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"".join(b"@" + n + b" c\n" + s + b"\n+\n" + q + b"\n" for n, s, q in recs))
    exe, ref = os.path.join(root, "bin", "rkmh"), os.path.join(data_dir, "all_pave_ref.fa.gz")
    rkp, rkp_nq = tmp_path / "r.rkp", tmp_path / "r.noq.rkp"
    _run([exe, "pack", "-f", str(fq), "-o", str(rkp), "--block-reads", "20000"])
    _run([exe, "pack", "-f", str(fq), "-o", str(rkp_nq), "--no-quals"])
    for cmd, flags in (("stream", []), ("stream", ["-M", "2", "-N", "3"]), ("stream", ["-k", "12", "-k", "20"]), ("filter", ["-N", "3"]), ("filter", ["-M", "2", "-N", "2"])):
        base = [exe, cmd, "-r", ref, "-s", "1000"] + (flags if "-k" in flags else ["-k", "16"] + flags)
        want = _run(base + ["-f", str(fq)])
        assert len(want) > 10000
        assert _run(base + ["-F", str(rkp)]) == want, (cmd, flags)
        assert _run(base + ["-F", str(rkp)], {"RKMH_PACKED_WORKERS": "1", "RKMH_PACKED_REGISTER": "0"}) == want, (cmd, flags)
        if cmd == "stream":
            assert _run(base + ["-F", str(rkp_nq)]) == want
            assert _run(base + ["-F", str(rkp), "-F", str(rkp_nq)]) == _run(base + ["-f", str(fq), "-f", str(fq)])
        else:   # without kept qualities filter prints empty quality lines, as for reads from FASTA
            got = _run(base + ["-F", str(rkp_nq)]).split(b"\n")
            w = want.split(b"\n")
            assert len(got) == len(w) and got[0::4] == w[0::4] and got[1::4] == w[1::4] and all(x == b"" for x in got[3::4])
    # what is not a packed file is refused with a message
    r = subprocess.run([exe, "stream", "-r", ref, "-F", str(fq)], capture_output=True)
    assert r.returncode == 1 and b"not a packed read file" in r.stderr and r.stdout == b""
    r = subprocess.run([exe, "stream", "-r", ref, "-F", str(rkp), "-f", str(fq)], capture_output=True)
    assert r.returncode == 1 and b"not both" in r.stderr


def test_packed_device_entry_equals_the_ascii_one(orc, data_dir):
    """rk_classify_batch_device_packed / rk_packed_slot_*: the rows of rk_classify_batch on the same reads, and the oracle's."""
    import ctypes as C
    import rkmh_amd
    from rkmh_amd import api
    lib = api.load_library()
    recs = _reads(data_dir, 8000, 2)
    refs = api.parse_files([os.path.join(data_dir, "all_pave_ref.fa.gz")])
    c = rkmh_amd.Context(0)
    try:
        c.set_references(refs["bases"], refs["offsets"], [16], 1000)
        qb, qo = orc.pack([s for _, s, _ in recs])
        qb = np.concatenate([qb, np.zeros(16, np.uint8)])
        want = c.classify(qb, qo)
        sk, ln = orc.sketch_refs(refs["bases"], refs["offsets"], [16], 1000, threads=8)
        assert (want == orc.classify_stream(qb, qo, [16], 1000, sk, ln, threads=8)).all()
        import torch
        nb = int(qo[-1])
        b2 = np.zeros((nb + 3) // 4 + 16, np.uint8)
        exc = np.zeros((nb + 1, 2), np.uint32)
        ne = lib.rk_packed_encode(qb.ctypes.data, nb, 0, b2.ctypes.data, exc.ctypes.data, nb + 1)
        assert ne > 0
        dev = "cuda:0"
        d_b2 = torch.from_numpy(b2).to(dev)
        d_off = torch.from_numpy(qo.astype(np.uint32).view(np.int32)).to(dev)
        d_exc = torch.from_numpy(exc[:ne].view(np.int32).copy()).to(dev)
        d_ascii = torch.zeros(16 * ((nb + 15) // 16) + 64, dtype=torch.uint8, device=dev)
        d_out = torch.zeros((len(recs), 4), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        rc = lib.rk_classify_batch_device_packed(c._h, d_b2.data_ptr(), d_off.data_ptr(), len(recs), nb, d_exc.data_ptr(), ne, d_ascii.data_ptr(), d_out.data_ptr(),
                                                 1800, lib.rk_ctx_stream(c._h))
        assert rc == 0, lib.rk_last_error()
        torch.cuda.synchronize()
        assert (d_out.cpu().numpy() == want).all()
        # the expanded bases are the originals, acgt folded
        got = bytes(d_ascii.cpu().numpy()[:nb])
        assert got == bytes(ch & 0xDF if (ch & 0xDF) in b"ACGT" else ch for ch in bytes(qb[:nb]))
    finally:
        c.close()
