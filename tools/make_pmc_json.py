"""Writes profiles/pmc_latest.json from a summary produced by tools/profile.sh (rocprofv3 --pmc passes of bench.py):
per-launch HBM traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md HBM section) and the VALU instruction
count of the fused kernel, stamped with the hash of the kernel sources they were measured on (rkmh_amd/stamp.py) and the
tile geometry -- bench.py drops the derived fields when the stamp no longer matches.
Usage: python3 tools/make_pmc_json.py profiles/<tag>_summary.txt [reads_per_launch]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rkmh_amd.stamp import kernel_source_stamp  # noqa: E402

src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
vals = {}
kernel = None
for line in open(src):
    m = re.match(r"^\s+(.*k_classify_(?:kmer|tile).*)\s(\w+)\s+n=(\d+)\s+mean=(\S+)", line)
    if m:
        kernel = kernel or m.group(1)
        vals[m.group(2)] = float(m.group(4))
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")
missing = [k for k in need if k not in vals]
if missing:
    sys.exit("make_pmc_json: %s has no %s for the classify kernel" % (src, ", ".join(missing)))
out = {
    "source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_* in separate passes, bench.py --steps 20)" % os.path.relpath(src, ROOT),
    "kernel": kernel,
    "kernel_source_stamp": kernel_source_stamp(),
    "reads_per_launch": n,
    "FETCH_SIZE_KB": vals["FETCH_SIZE"],
    "WRITE_SIZE_KB": vals["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); narrow random accesses are uncalibrated and "
                  "Infinity-Cache hits are counted, so this is an upper bound on HBM bytes",
    "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024),
    "valu_insts_per_launch": vals["SQ_INSTS_VALU"],
    "salu_insts_per_launch": vals.get("SQ_INSTS_SALU"),
    "lds_insts_per_launch": vals.get("SQ_INSTS_LDS"),
    "gui_active_cycles_per_launch": vals.get("GRBM_GUI_ACTIVE"),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
