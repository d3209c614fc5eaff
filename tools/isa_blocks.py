"""Basic-block instruction statistics of one kernel in a hipcc -save-temps .s file (VALU/SALU/LDS/VMEM counts per block)."""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().split(":")[0] == sym or l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks = []; cur = ("entry", [])
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur); cur = (t.split(":")[0], [])
    elif t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t.split(";")[0].strip())
blocks.append(cur)
def cls(i):
    op = i.split()[0]
    for p, c in (("v_", "V"), ("s_", "S"), ("ds_", "L"), ("global_", "G"), ("buffer_", "G"), ("flat_", "G"), ("scratch_", "X")):
        if op.startswith(p): return c
    return "O"
tot = {}
for name, ins in blocks:
    c = {}
    for i in ins:
        c[cls(i)] = c.get(cls(i), 0) + 1; tot[cls(i)] = tot.get(cls(i), 0) + 1
    mul = sum(1 for i in ins if re.match(r"v_mul_(lo|hi)_u32|v_mad_u64_u32|v_mul_u32_u24|v_mad_u32_u24", i))
    br = [i for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
    print("%-10s n=%4d V=%4d S=%4d L=%3d G=%2d mul=%3d  %s" % (name, len(ins), c.get("V", 0), c.get("S", 0), c.get("L", 0), c.get("G", 0), mul, " | ".join(br)))
print("total", tot)
