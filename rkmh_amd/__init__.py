"""rkmh_amd -- MI355X-native MinHash read classification (the classify/stream hot path of edawson/rkmh).

The product is librkmh_amd.so (hand-written HIP kernels for gfx950 behind the C ABI of include/rkmh_amd.h) and
the `rkmh` command line built on it.  This Python package is plumbing: a ctypes binding of that C ABI
(`rkmh_amd.api`), the synthetic-workload generator of SURVEY.md section 8(d) (`rkmh_amd.synth`) and the
one-process-per-GPU sharding over torch.distributed / RCCL (`rkmh_amd.dist`).

There is no CPU fallback: without the built library or without a GPU every compute call raises.
"""
import os as _os

# A stream per front-end worker: the runtime maps streams onto four hardware queues unless told otherwise, and kernels of two streams
# on one queue run one after the other (profiles/r05_gz.txt).  Read when the HIP runtime starts -- without effect if it already has.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .api import Context, Counter, RkmhError, library_path, load_library, parse_files  # noqa: F401

__all__ = ["Context", "Counter", "RkmhError", "library_path", "load_library", "parse_files"]
