"""The drop-in's host side at a given size: kmc_hip -r on ONE FASTQ of [Gbp] under different environments (reader threads, pinned-pool size, allocator tunables ...);
"2nd stage", the process's wall time, the worker report, the host-boundary phase times (kmc_hip_host_boundary_times) and the timeline of every run.
usage: python tools/e2e_reader_sweep.py [Gbp] ['<json list of env dicts>']"""
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
envs = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "16", "KMC_HIP_PINNED_POOL_MB": "4096"}]
reads = int(gbp * 1e9 / 150)
hip = os.path.join(ROOT, "kmc_amd", "bin", "kmc_hip")
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    fq = os.path.join(td, "l.fq")
    capi.synth_fastq(fq, seed=2027, genome_len=reads * 5, n_reads=reads)
    for env in envs:
        tmp = os.path.join(td, "tmp")
        os.makedirs(tmp, exist_ok=True)
        t = time.time()
        r = subprocess.run([hip, "-k27", "-t16", "-m512", "-r", "-sr16", "-hp", fq, os.path.join(td, "db"), tmp], capture_output=True, text=True,
                           env=dict(os.environ, KMC_HIP_LIB=capi.lib_path(), KMC_HIP_VERBOSE="1", **env))
        wall = time.time() - t
        s1 = re.search(r"1st stage:\s*([0-9.eE+-]+)s", r.stdout)
        s2 = re.search(r"2nd stage:\s*([0-9.eE+-]+)s", r.stdout)
        tot = re.search(r"Total\s*:\s*([0-9.eE+-]+)s", r.stdout)
        uniq = re.search(r"No\. of unique k-mers\s*:\s*(\d+)", r.stdout)
        rep = [ln for ln in r.stderr.splitlines() if "reader: admit" in ln]
        hb = [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip host boundary]")]
        tl = [ln for ln in r.stderr.splitlines() if ln.startswith("[kmc_hip timeline]")]
        print(json.dumps({"env": env, "rc": r.returncode, "stage1_s": float(s1.group(1)) if s1 else None, "stage2_s": float(s2.group(1)) if s2 else None,
                          "kmc_total_s": float(tot.group(1)) if tot else None, "process_wall_s": round(wall, 2), "unique": int(uniq.group(1)) if uniq else None,
                          "report": rep[0][rep[0].index("reader:"):][:330] if rep else None, "host_boundary": hb[0][24:] if hb else None,
                          "timeline": tl[0][60:600] if tl else None, "stderr_tail": r.stderr[-300:] if r.returncode else None}), flush=True)
