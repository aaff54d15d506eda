#!/bin/bash
# round 6, session s: the drop-in with the allocator tunables + registered pinned pool as defaults: 8 Gbp sweep, then the 30 Gbp run (reference beside it)
OUT=gpurun_out/r06s; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ENVS='[{}, {"KMC_HIP_TUNE_MALLOC": "0"}, {"KMC_HIP_TUNE_MALLOC": "0", "KMC_HIP_PINNED_POOL_MB": "1024"}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "4"}, {"KMC_HIP_READERS": "2"}, {"KMC_HIP_WORKER_GROUP": "1"}, {"KMC_HIP_PINNED_POOL_MB": "0"}, {}]'
timeout 900 python tools/e2e_reader_sweep.py 8 "$ENVS" > $OUT/e2e_sweep_8gbp.jsonl 2> $OUT/e2e_sweep_8gbp.err; python - <<'PY'
import json
for ln in open("gpurun_out/r06s/e2e_sweep_8gbp.jsonl"):
    d=json.loads(ln); print(d["env"], "rc", d["rc"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("host_boundary") or "")[:230], "|", (d.get("timeline") or "")[88:330], (d.get("stderr_tail") or "")[-200:])
PY
free -g | head -2
timeout 1200 python tools/e2e_large_run.py 30 > $OUT/e2e_large_30gbp.json 2> $OUT/e2e_large_30gbp.err; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r06s/e2e_large_30gbp.json")); print({k:d[k] for k in d if k in ("ref_stage1_s","ref_stage2_s","hip_stage1_s","hip_stage2_s","speedup","hip_Gkmers_per_s","stats_equal","worker_report")}); print(d.get("timeline")); print(d.get("device_resident_on_reference_bins"))
except Exception as e: print("30gbp", e); print(open("gpurun_out/r06s/e2e_large_30gbp.err").read()[-600:])
PY
