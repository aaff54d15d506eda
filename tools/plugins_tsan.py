#!/usr/bin/env python3
"""The worker / reader / completer / splitter plug-ins (kmc_amd/host/*.h) inside the reference pipeline under ThreadSanitizer: builds
oracle/_ref-style binaries with -fsanitize=thread into /tmp/ref_tsan (oracle/Makefile with OUT/CXX overridden; engines = the oracles, no GPU) and
runs kmc_oracle_all on a small FASTQ with several thread configurations. Prints the number of reports per run and where they point; a clean run
prints zeros. Reports inside /root/reference code are the reference's own business and are listed separately."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import synth  # noqa: E402

OUT = "/tmp/ref_tsan"
subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), f"OUT={OUT}", f"OBJ={OUT}/obj", "CXX=g++ -fsanitize=thread -g", "CC=gcc -fsanitize=thread -g",
                       f"{OUT}/kmc_oracle_all"])
td = tempfile.mkdtemp(dir="/dev/shm")
fq = os.path.join(td, "r.fq")
synth.make_fastq(fq, seed=11, genome_len=300_000, n_reads=30_000, read_len=150)
RUNS = [(["-k27", "-sf2", "-sp4", "-sr6"], {}), (["-k27", "-sf1", "-sp3", "-sr12", "-r"], {"KMC_HIP_READERS": "3"}), (["-k55", "-sf2", "-sp2", "-sr4"], {}),
        (["-k27", "-b", "-sf2", "-sp6", "-sr2"], {"KMC_HIP_WRITERS": "1"}), (["-k27", "-sf2", "-sp8", "-sr16"], {"KMC_HIP_READERS": "16"})]
total = 0
for flags, extra in RUNS:
    t = os.path.join(td, "t")
    shutil.rmtree(t, ignore_errors=True)
    os.makedirs(t)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=3")
    env.update({"KMC_HIP_READERS": "8", "KMC_HIP_WRITERS": "4"})
    env.update(extra)
    r = subprocess.run([os.path.join(OUT, "kmc_oracle_all"), *flags, "-ci1", "-m2", fq, os.path.join(td, "db"), t], capture_output=True, text=True, env=env)
    where = collections.Counter()
    for block in r.stderr.split("=================="):
        if "WARNING: ThreadSanitizer" not in block:
            continue
        frames = re.findall(r"#\d+ .*? (/[^ :]+):(\d+)", block)
        ours = [f for f in frames if "/kmc_amd/" in f[0] or "/oracle/" in f[0]][:2]
        where[tuple(os.path.basename(f[0]) + ":" + f[1] for f in ours) or ("reference only",)] += 1
    n = sum(where.values())
    total += n
    print(flags, extra, "rc", r.returncode, "reports", n, dict(where))
# the product binary with the dlopen loaders (hip_loader.cpp, hip_split_loader.cpp: eager-initialisation thread, call_once, slot mutexes) over the mock library
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu  # noqa: E402

mock = emu.build_mock()
subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "kmc_amd", "host"), f"OUT={OUT}/bin", f"OBJ={OUT}/bin/obj", f"ROBJ={OUT}/obj",
                       "CXX=g++ -fsanitize=thread -g", "CC=gcc -fsanitize=thread -g", f"{OUT}/bin/kmc_hip_s1"])
small = os.path.join(td, "small.fq")
synth.make_fastq(small, seed=11, genome_len=50_000, n_reads=3_000, read_len=150)
for flags, extra in [(["-k27", "-sf2", "-sp3", "-sr4"], {}), (["-k27", "-sf1", "-sp2", "-sr6"], {"KMC_HIP_EAGER_INIT": "0", "KMC_HIP_DEVICES": "0,1"})]:
    t = os.path.join(td, "t")
    shutil.rmtree(t, ignore_errors=True)
    os.makedirs(t)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=3", KMC_HIP_LIB=mock)
    env.update(extra)
    r = subprocess.run([os.path.join(OUT, "bin", "kmc_hip_s1"), *flags, "-ci1", "-m2", small, os.path.join(td, "db"), t], capture_output=True, text=True, env=env)
    n = r.stderr.count("WARNING: ThreadSanitizer")
    total += n
    print("kmc_hip_s1 over the mock library", flags, extra, "rc", r.returncode, "reports", n)
shutil.rmtree(td)
print("total reports:", total)
