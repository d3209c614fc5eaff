"""Identity of the fused kernel's sources: measurements kept beside the code (profiles/pmc_latest.json) carry this stamp,
and bench.py only quotes them while it still matches the sources that were built."""
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KERNEL_SOURCES = ("rk_classify.hip", "rk_device.hpp")  # everything k_classify_tile is compiled from but shared structs


def kernel_source_stamp() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]
