#!/bin/bash
# round 6, session v: the 30 Gbp run (reference beside the drop-in, RAM-only mode), default allocator as in round 5, with the slot slabs and the reorder buffer
OUT=gpurun_out/r06v; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
free -g | head -2
timeout 1200 python tools/e2e_large_run.py 30 > $OUT/e2e_large_30gbp.json 2> $OUT/e2e_large_30gbp.err; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r06v/e2e_large_30gbp.json")); print({k:d[k] for k in d if k in ("ref_stage1_s","ref_stage2_s","hip_stage1_s","hip_stage2_s","speedup","hip_Gkmers_per_s","stats_equal","worker_report")}); print(d.get("timeline")); print(d.get("device_resident_on_reference_bins"))
except Exception as e: print("30gbp", e); print(open("gpurun_out/r06v/e2e_large_30gbp.err").read()[-600:])
PY
free -g | head -2
