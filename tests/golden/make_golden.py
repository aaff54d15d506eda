#!/usr/bin/env python3
"""Regenerates tests/golden/*.bins — run HERE (needs /root/reference and oracle/_ref built by oracle/Makefile).

For each case: the REAL reference stage 1 (inside oracle/_ref/kmc_oracle = reference pipeline + oracle sorter)
cuts the input into signature bins; every bin's image, pack list and the stage-2 outputs are teed to a dump file
($KMC_BIN_DUMP). The whole-database bytes of that run are first checked to be identical to the unmodified reference
(oracle/_ref/kmc -sr1), so each dumped (image -> out, lut, tallies) triple is a reference-grade golden vector.
A few bins per case are kept to stay small.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import golden_io  # noqa: E402
from kmc_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
DATA = "/root/reference/tests/kmc_CLI/data"

CASES = [
    # name, input, kmc flags, how many non-empty bins to keep (the largest and the smallest ones) + 1 empty bin if present
    ("single_read_k28", os.path.join(DATA, "single_read.fq"), ["-k28", "-ci1"], 8),
    # (issue-180's k=5 case takes KMC's small-k path, kmc.h:677-760: no bins, out of scope; its subject — a k-mer
    #  equal to its own reverse complement — is covered on the bin path by the even-k case c1_k14 below.)
    ("c1_k14_ci1", "C1", ["-k14", "-ci1"], 2),
    ("c1_k27", "C1", ["-k27"], 4),
    ("c1_k27_b_ci1_cs3", "C1", ["-k27", "-b", "-ci1", "-cs3"], 2),
    ("c1_k55", "C1", ["-k55"], 3),
    ("c1_k127", "C1", ["-k127"], 2),
    ("c1_k32_cx", "C1", ["-k32", "-ci1", "-cx3"], 2),
    ("c1_k21_n64", "C1", ["-k21", "-n64"], 1),
]


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def main():
    with tempfile.TemporaryDirectory() as td:
        c1 = os.path.join(td, "c1.fq")
        synth.make_fastq(c1, **synth.CONFIGS["C1"])
        for name, inp, flags, keep in CASES:
            inp = c1 if inp == "C1" else inp
            dump = os.path.join(td, name + ".dump")
            for exe, out in (("kmc", "ref"), ("kmc_oracle", "orc")):
                tmp = os.path.join(td, f"tmp_{out}")
                os.makedirs(tmp, exist_ok=True)
                env = dict(os.environ)
                if exe == "kmc_oracle":
                    env["KMC_BIN_DUMP"] = dump
                subprocess.check_call([os.path.join(REF, exe), *flags, "-sr1", inp, os.path.join(td, out), tmp], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for ext in (".kmc_pre", ".kmc_suf"):
                assert md5(os.path.join(td, "ref" + ext)) == md5(os.path.join(td, "orc" + ext)), f"{name}: oracle DB != reference DB"
            bins = list(golden_io.read_bins(dump))
            nonempty = sorted([b for b in bins if b["n_rec"]], key=lambda b: -b["n_rec"])
            empty = [b for b in bins if not b["n_rec"]][:1]
            sel = nonempty[:1] + nonempty[len(nonempty) - (keep - 1):][::-1] * (keep > 1) + empty  # largest + the smallest ones
            golden_io.write_bins(os.path.join(HERE, name + ".bins"), sel)
            print(f"{name}: {len(bins)} bins dumped, kept {len(sel)}; sizes {[b['size'] for b in sel]}")
            os.remove(dump)


if __name__ == "__main__":
    main()
