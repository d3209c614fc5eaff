"""Generates tests/golden/murmur3_kat.json: known-answer vectors for MurmurHash3_x64_128.

The vectors come from an implementation INDEPENDENT of both this repo and /root/reference:
the public-domain Appleby source that scikit-learn vendors in this container
(<site-packages>/sklearn/utils/src/MurmurHash3.cpp), compiled in a temp dir.  They pin the murmur
core of the oracle and of the HIP kernels (SURVEY.md section 8c "Known-answer tests"); they do NOT pin
mkmh's fold / shredding (parity unpinned).  Run: python tests/golden/gen_murmur3_kat.py
"""
import json, os, random, subprocess, tempfile
import sklearn

src_dir = os.path.join(os.path.dirname(sklearn.__file__), "utils", "src")
drv = r'''
#include "MurmurHash3.h"
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
int main(int argc, char** argv){
  // argv: seed hexkey
  uint32_t seed = (uint32_t)strtoul(argv[1], 0, 10);
  const char* hex = argv[2]; int n = strlen(hex)/2; unsigned char buf[4096];
  for (int i=0;i<n;i++){ unsigned v; sscanf(hex+2*i, "%2x", &v); buf[i]=(unsigned char)v; }
  uint64_t out[2]; MurmurHash3_x64_128(buf, n, seed, out);
  printf("%llu %llu\n", (unsigned long long)out[0], (unsigned long long)out[1]);
  return 0; }
'''
rng = random.Random(20261001)
cases = [(0, b"hello"), (42, b"ACGTACGTACGTACGT"), (42, b""), (0, b""), (42, b"A"), (42, b"ACGTACGTACGT"),
         (42, b"ACGTACGTACGTACGTACGT"), (42, b"TTTTTTTTTTTTTTTT"), (42, b"GATTACAGATTACAGATTACAGATTACAGATTACA")]
for L in list(range(1, 40)) + [48, 63, 64, 65, 100]:
    cases.append((42, bytes(rng.choice(b"ACGT") for _ in range(L))))
for L in (7, 16, 31, 33):
    cases.append((rng.randrange(1 << 32), bytes(rng.randrange(256) for _ in range(L))))
with tempfile.TemporaryDirectory() as td:
    open(os.path.join(td, "drv.cpp"), "w").write(drv)
    exe = os.path.join(td, "kat")
    subprocess.check_call(["g++", "-O1", "-I", src_dir, os.path.join(td, "drv.cpp"),
                           os.path.join(src_dir, "MurmurHash3.cpp"), "-o", exe])
    vec = []
    for seed, key in cases:
        o = subprocess.check_output([exe, str(seed), key.hex() if key else ""]).split() if key else \
            subprocess.check_output([exe, str(seed), ""]).split()
        vec.append({"seed": seed, "key_hex": key.hex(), "h1": int(o[0]), "h2": int(o[1])})
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "murmur3_kat.json")
json.dump({"source": "sklearn/utils/src/MurmurHash3.cpp (Appleby, public domain), sklearn " + sklearn.__version__,
           "vectors": vec}, open(out, "w"), indent=0)
print("wrote", out, len(vec))
