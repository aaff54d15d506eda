set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess
sys.path.insert(0,'.')
from kmc_amd import synth
fq='/dev/shm/r.fq'
synth.make_fastq(fq, seed=2026, genome_len=66_000_000, n_reads=13_300_000, read_len=150)
for exe, extra in (("kmc_amd/bin/kmc_hip", {}), ("kmc_amd/bin/kmc_hip_s1", {}), ("kmc_amd/bin/kmc_hip_s1", {"KMC_HIP_SPLITTER_REF":"1"})):
    os.makedirs('/dev/shm/t', exist_ok=True)
    env=dict(os.environ, KMC_HIP_VERBOSE="1", **extra)
    r=subprocess.run([exe,"-k27","-t128","-m128","-sr16",fq,"/dev/shm/db","/dev/shm/t"],capture_output=True,text=True,env=env)
    print("=====",exe,extra)
    print("\n".join(l for l in r.stdout.splitlines() if "stage" in l.lower() or "time" in l.lower()))
    err=[l for l in r.stderr.splitlines() if l.startswith("[kmc_hip")]
    print("\n".join(l[:600] for l in err if "worker:" not in l))
PY
