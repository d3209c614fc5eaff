import json
import os

import numpy as np


def load_refs_reads(orc, data_dir, ref_file, reads_file):
    r = orc.kseq_parse_file(os.path.join(data_dir, ref_file))
    q = orc.kseq_parse_file(os.path.join(data_dir, reads_file))
    return r, q


def golden(golden_dir, tag):
    return json.load(open(os.path.join(golden_dir, "classify_%s.json" % tag)))


def rand_dna(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n).tolist())
