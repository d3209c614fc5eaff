"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

The classify path shards by reads -- each rank owns a contiguous block of the global read set and never
talks to the others after start-up.  The only exchanges are:
  * broadcast of the reference sketches uint64[R][S] + int32 lens[R] from the rank that sketched them;
  * (-M path only) an all-reduce(sum) of the HASHTCounter table between pass 1 and pass 2.
"""
import os

import numpy as np


def env_rank():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RKMH_ONE_DEVICE"):   # plumbing tests: several ranks share GPU 0 (use with RKMH_DIST_BACKEND=gloo)
        local = 0
    return int(os.environ.get("RANK", "0")), local, int(os.environ.get("WORLD_SIZE", "1"))


def _collectives_on():
    """The collectives run when there is more than one rank -- or with a single rank when RKMH_DIST_FORCE=1, which exists so that the
    RCCL code path (device tensors, nccl backend) can be executed on a one-GPU box (tests/test_gpu_parity.py)."""
    import torch.distributed as dist
    return dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("RKMH_DIST_FORCE")))


def init(backend=None):
    import torch
    import torch.distributed as dist
    rank, local, world = env_rank()
    if (world > 1 or os.environ.get("RKMH_DIST_FORCE")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("RKMH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_bounds(n_total, rank, world):
    """Contiguous block [lo, hi) of rank (GPU g gets reads [g*N/G, (g+1)*N/G))."""
    return n_total * rank // world, n_total * (rank + 1) // world


def broadcast_sketches(sketches, lens, nref, sketch_size, src=0, device=None):
    """Rank `src` passes (sketches [R,S] uint64, lens [R] int32); the others pass None. Returns numpy arrays."""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return sketches, lens
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    if dist.get_rank() == src:
        t_sk = torch.from_numpy(np.ascontiguousarray(sketches).view(np.int64)).to(dev)
        t_ln = torch.from_numpy(np.ascontiguousarray(lens)).to(dev)
    else:
        t_sk = torch.empty((nref, sketch_size), dtype=torch.int64, device=dev)
        t_ln = torch.empty((nref,), dtype=torch.int32, device=dev)
    dist.broadcast(t_sk, src=src)
    dist.broadcast(t_ln, src=src)
    return t_sk.cpu().numpy().view(np.uint64), t_ln.cpu().numpy()


def allreduce_counter(t_counts):
    """Sum the per-rank HASHTCounter tables in place (torch int32 tensor that the rk_counter wraps)."""
    import torch.distributed as dist
    if _collectives_on():
        dist.all_reduce(t_counts, op=dist.ReduceOp.SUM)
    return t_counts


def gather_rows(rows, dst=0):
    """Gather per-rank result blocks (numpy [n_i,4] int32) on `dst` in rank order (= global read order).
    Tensor collectives only (all_gather of the block sizes, gather of blocks padded to the largest): no pickling of
    gigabyte-sized results, and the same code under RCCL (device tensors) and gloo (host tensors)."""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return rows
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
    t_n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, t_n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    if rows.shape[0]:
        buf[: rows.shape[0]] = torch.from_numpy(rows).to(dev)
    if rank == dst:
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, outs, dst=dst)
        return np.concatenate([outs[r][: sizes[r]].cpu().numpy() for r in range(world)], axis=0)
    dist.gather(buf, None, dst=dst)
    return None


def all_true(flag):
    """Logical AND of a per-rank boolean (MIN all-reduce): does every rank agree that a shortcut is safe?"""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return bool(flag)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def gather_bytes(data, dst=0):
    """Per-rank byte strings concatenated on `dst` in rank order (formatted output lines of the ranks' read shards); None elsewhere."""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return bytes(data)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t_n = torch.tensor([len(data)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, t_n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(data):
        buf[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    if rank == dst:
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, outs, dst=dst)
        return b"".join(outs[r][: sizes[r]].cpu().numpy().tobytes() for r in range(world))
    dist.gather(buf, None, dst=dst)
    return None


def all_gather_int(value):
    """Every rank's integer, in rank order (the byte counts that place the ranks' output blocks in a shared file)."""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return [int(value)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    outs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [int(x.item()) for x in outs]


def barrier():
    import torch.distributed as dist
    if _collectives_on():
        dist.barrier()


def broadcast_bytes(data, src=0):
    """A byte string from rank `src` to every rank (the reference names, so that only one rank reads the reference files)."""
    import torch
    import torch.distributed as dist
    if not _collectives_on():
        return bytes(data)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t_n = torch.tensor([len(data) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(t_n, src=src)
    n = int(t_n.item())
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8).to(dev)
    else:
        buf = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return buf[:n].cpu().numpy().tobytes()
