"""The drop-in's host side at the configs[2] size: kmc_hip -r on ONE 30 Gbp FASTQ with different numbers of reader threads / pinned-pool sizes; "2nd stage" and the worker report."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmc_amd import capi  # noqa: E402

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
reads = int(gbp * 1e9 / 150)
hip = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip")
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    fq = os.path.join(td, "l.fq")
    capi.synth_fastq(fq, seed=2027, genome_len=reads * 5, n_reads=reads)
    for env in ({}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "24"}, {"KMC_HIP_READERS": "16", "KMC_HIP_PINNED_POOL_MB": "4096"}, {"KMC_HIP_READERS": "16", "KMC_HIP_WRITERS": "8"}):
        tmp = os.path.join(td, "tmp")
        os.makedirs(tmp, exist_ok=True)
        t = time.time()
        r = subprocess.run([hip, "-k27", "-t16", "-m512", "-r", "-sr16", "-hp", fq, os.path.join(td, "db"), tmp], capture_output=True, text=True,
                           env=dict(os.environ, KMC_HIP_LIB=capi.lib_path(), KMC_HIP_VERBOSE="1", **env))
        s2 = re.search(r"2nd stage:\s*([0-9.eE+-]+)s", r.stdout)
        rep = [ln for ln in r.stderr.splitlines() if "reader: admit" in ln]
        tl = [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip timeline]")]
        print(json.dumps({"env": env, "rc": r.returncode, "stage2_s": float(s2.group(1)) if s2 else None, "wall_s": round(time.time() - t, 1),
                          "report": rep[0][rep[0].index("reader:"):][:330] if rep else None, "timeline": tl[0][60:560] if tl else None}), flush=True)
