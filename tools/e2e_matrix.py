#!/usr/bin/env python3
"""Where does the drop-in's stage 2 spend its time? Runs kmc_amd/bin/kmc_hip (worker + reader plug-ins), kmc_hip_sr (worker plug-in,
reference reader) and the unmodified reference on ONE FASTQ (the 2 Gbp sample of bench.py) for several -sr / KMC_HIP_READERS /
-r settings and prints one JSON line per run: stage times, the five statistics, the plug-ins' KMC_HIP_VERBOSE report."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 13_300_000
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 66_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 27
base = "/dev/shm" if shutil.disk_usage("/dev/shm").free > reads * 316 * 3 else "/tmp"
REF = os.path.join(ROOT, "oracle", "_ref")
with tempfile.TemporaryDirectory(dir=base) as td:
    fq = os.path.join(td, "s.fq")
    t = time.time()
    capi.synth_fastq(fq, seed=2026, genome_len=genome, n_reads=reads)
    print(json.dumps({"fastq_s": time.time() - t, "reads": reads, "genome": genome, "k": k}), flush=True)
    runs = [("kmc", ["-t128", "-m128"], {}),
            ("kmc", ["-t16", "-m128"], {}),
            ("kmc", ["-t32", "-m128"], {}),
            ("kmc_hip", ["-t32", "-m128", "-sr16"], {}),
            ("kmc_hip", ["-t32", "-m128", "-sr16"], {"KMC_HIP_COMPLETER": "ref"}),
            ("kmc_hip", ["-t32", "-m128", "-sr12"], {"KMC_HIP_READERS": "4", "KMC_HIP_WRITERS": "2"}),
            ("kmc_hip", ["-t128", "-m128", "-sr16"], {}),
            ("kmc_hip", ["-t128", "-m128", "-sr16"], {"KMC_HIP_READERS": "1"}),
            ("kmc_hip", ["-t128", "-m128", "-sr16"], {"KMC_HIP_READERS": "16"}),
            ("kmc_hip", ["-t128", "-m128", "-sr8"], {}),
            ("kmc_hip", ["-t128", "-m128", "-sr32"], {}),
            ("kmc_hip", ["-t128", "-m128", "-sr16", "-r"], {}),
            ("kmc_hip", ["-t128", "-m16", "-sr16"], {}),
            ("kmc_hip_sr", ["-t128", "-m128", "-sr16"], {}),
            ("kmc", ["-t128", "-m128", "-r"], {}),
            # indices 15..19: stage 1 on the GPU too (kmc_hip_s1)
            ("kmc_hip_s1", ["-t32", "-m128", "-sr16"], {}),
            ("kmc_hip_s1", ["-t128", "-m128", "-sr16"], {}),
            ("kmc_hip_s1", ["-t128", "-m128", "-sr16", "-sp16", "-sf4"], {}),
            ("kmc_hip_s1", ["-t128", "-m128", "-sr16", "-sp32", "-sf8"], {}),
            ("kmc_hip_s1", ["-t32", "-m128", "-sr16"], {"KMC_HIP_SPLITTER_REF": "1"})]
    if len(sys.argv) > 4:  # a subset of the runs, by index
        runs = [runs[int(i)] for i in sys.argv[4].split(",")]
    for exe, flags, env in runs:
        p = os.path.join(ROOT, "kmc_amd", "bin", exe) if exe.startswith("kmc_hip") else os.path.join(REF, exe)
        if not os.path.exists(p):
            continue
        tmp = os.path.join(td, "tmp")
        os.makedirs(tmp, exist_ok=True)
        e = dict(os.environ, KMC_HIP_LIB=capi.lib_path(), KMC_HIP_VERBOSE="1", **env)
        t = time.time()
        r = subprocess.run([p, f"-k{k}", *flags, fq, os.path.join(td, "db"), tmp], capture_output=True, text=True, env=e)
        wall = time.time() - t
        shutil.rmtree(tmp, ignore_errors=True)
        m1 = re.search(r"1st stage:\s*([0-9.eE+-]+)s", r.stdout)
        m2 = re.search(r"2nd stage:\s*([0-9.eE+-]+)s", r.stdout)
        tot = re.search(r"Total no\. of k-mers\s*:\s*(\d+)", r.stdout)
        uq = re.search(r"No\. of unique k-mers\s*:\s*(\d+)", r.stdout)
        print(json.dumps({"exe": exe, "flags": flags, "env": env, "rc": r.returncode, "stage1_s": float(m1.group(1)) if m1 else None,
                          "stage2_s": float(m2.group(1)) if m2 else None, "wall_s": wall, "total": int(tot.group(1)) if tot else None,
                          "unique": int(uq.group(1)) if uq else None, "report": [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip")],
                          "err": r.stderr[-300:] if r.returncode else ""}), flush=True)
