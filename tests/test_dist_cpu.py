"""CPU-only, world_size 2 over gloo: the multi-GPU plumbing of the classify path (rkmh_amd/dist.py) --
shard bounds, the reference-sketch broadcast, the -M counter all-reduce and the ordered gather of result rows."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rkmh_amd import dist as rdist
    r, lr, w = rdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    # 1. contiguous read shards cover [0, N) exactly once
    N = 1000003
    lo, hi = rdist.shard_bounds(N, rank, world)
    # 2. sketches built on rank 0 reach everyone bit-exact (uint64 incl. values >= 2^63)
    R, S = 7, 50
    rng = np.random.default_rng(5)
    sk0 = rng.integers(0, np.iinfo(np.uint64).max, size=(R, S), dtype=np.uint64)
    ln0 = rng.integers(0, S + 1, size=R).astype(np.int32)
    sk, ln = rdist.broadcast_sketches(sk0 if rank == 0 else None, ln0 if rank == 0 else None, R, S, src=0)
    ok_b = bool((sk == sk0).all() and (ln == ln0).all() and sk.dtype == np.uint64)
    # 3. -M: per-rank counter tables sum to the global table
    t = torch.full((1000,), rank + 1, dtype=torch.int32)
    rdist.allreduce_counter(t)
    ok_c = bool((t == sum(range(1, world + 1))).all())
    # 4. result rows come back in global read order
    rows = np.full((hi - lo, 4), rank, dtype=np.int32)
    rows[:, 1] = np.arange(lo, hi)
    allrows = rdist.gather_rows(rows, dst=0)
    ok_g = True
    if rank == 0:
        ok_g = allrows.shape == (N, 4) and bool((allrows[:, 1] == np.arange(N)).all())
    else:
        ok_g = allrows is None
    # 5. formatted text of the ranks' shards comes back in rank order; a shortcut is taken only if every rank agrees
    text = rdist.gather_bytes((b"rank%d\n" % rank) * (rank + 2), dst=0)
    ok_g = ok_g and (text == b"".join((b"rank%d\n" % r) * (r + 2) for r in range(world)) if rank == 0 else text is None)
    ok_g = ok_g and rdist.gather_bytes(b"" if rank else b"only rank 0", dst=0) == (b"only rank 0" if rank == 0 else None)
    ok_g = ok_g and rdist.all_true(True) is True and rdist.all_true(rank != 1) is False
    # the reference names travel from rank 0; byte counts are all-gathered; a barrier is a barrier
    ok_g = ok_g and rdist.broadcast_bytes(b"a\0bc\0" * 1000 if rank == 0 else None, src=0) == b"a\0bc\0" * 1000
    ok_g = ok_g and rdist.broadcast_bytes(b"" if rank == 0 else None, src=0) == b""
    ok_g = ok_g and rdist.all_gather_int(10 + rank) == [10 + r for r in range(world)]
    rdist.barrier()
    # 6. the ranks' output blocks land in rank order without travelling: pwrite at final offsets into a regular file, in turn into a pipe
    out_path = os.environ.get("RKMH_TEST_OUT")
    if out_path:
        from rkmh_amd import cli
        fd = os.open(out_path, os.O_WRONLY)
        sink = cli._RankOutput(fd, rank, world)
        ok_g = ok_g and sink.regular
        for f in range(2):      # two input files: every file's text in rank order, file after file
            sink.pieces_of_a_file([b"file%d rank%d piece%d\n" % (f, rank, j) * (50 + 7 * rank) for j in range(3)])
        sink.finish()
        os.close(fd)
        fifo = out_path + ".fifo"
        fd = os.open(fifo, os.O_WRONLY)
        sink = cli._RankOutput(fd, rank, world)
        ok_g = ok_g and not sink.regular
        sink.pieces_of_a_file([b"pipe rank%d\n" % rank * 1000])
        os.close(fd)
        # every rank its OWN stdout (torchrun --redirects): nothing may be written at offsets of a file the others do not share --
        # rank 0 gathers the text and writes it alone
        fd = os.open(out_path + ".rank%d" % rank, os.O_WRONLY | os.O_CREAT, 0o644)
        sink = cli._RankOutput(fd, rank, world)
        ok_g = ok_g and sink.regular and not sink.shared
        for f in range(2):
            sink.pieces_of_a_file([b"own file%d rank%d piece%d\n" % (f, rank, j) * (20 + 3 * rank) for j in range(2)])
        sink.finish()
        os.close(fd)
    q.put((rank, lo, hi, ok_b, ok_c, ok_g))
    torch.distributed.destroy_process_group()


def test_two_rank_plumbing_over_gloo(tmp_path):
    world = 2
    port = _free_port()
    out = tmp_path / "shared_out.txt"
    out.write_bytes(b"")
    os.mkfifo(str(out) + ".fifo")
    os.environ["RKMH_TEST_OUT"] = str(out)
    import threading
    piped = []
    reader = threading.Thread(target=lambda: piped.append(open(str(out) + ".fifo", "rb").read()))
    reader.start()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 1000003
    reader.join(timeout=60)
    del os.environ["RKMH_TEST_OUT"]
    want = b"".join(b"file%d rank%d piece%d\n" % (f, r, j) * (50 + 7 * r) for f in range(2) for r in range(world) for j in range(3))
    assert out.read_bytes() == want
    assert piped == [b"".join(b"pipe rank%d\n" % r * 1000 for r in range(world))]
    own = b"".join(b"own file%d rank%d piece%d\n" % (f, r, j) * (20 + 3 * r) for f in range(2) for r in range(world) for j in range(2))
    assert open(str(out) + ".rank0", "rb").read() == own and open(str(out) + ".rank1", "rb").read() == b""
    for r in res:
        assert r[3] and r[4] and r[5], r


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from rkmh_amd import dist as rdist
    for n in (0, 1, 7, 100, 1000003):
        for w in (1, 2, 3, 8):
            b = [rdist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
