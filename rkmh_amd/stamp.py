"""Identity of the hot kernels' sources: measurements kept beside the code (profiles/pmc_latest.json) carry this stamp,
and bench.py only quotes them while it still matches the sources that were built."""
import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# everything that shapes the profiled kernel's instruction counts and traffic: the kernels, the shared device code, the index /
# filter / map sizing on the host side (rk_api.hip), and the build flags
KERNEL_SOURCES = ("rk_kmer.hip", "rk_classify.hip", "rk_device.hpp", "rk_kernels.hip", "rk_kernels.hpp", "rk_api.hip")


def kernel_source_stamp() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    with open(os.path.join(_ROOT, "Makefile"), "rb") as f:
        h.update(b"Makefile\0" + f.read() + b"\0")
    h.update(os.environ.get("EXTRA_HIPFLAGS", "").encode())
    return h.hexdigest()[:16]
