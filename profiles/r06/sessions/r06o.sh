#!/bin/bash
# round 6, session o: (1) new / changed -m gpu cases on the trimmed library; (2) KMC_HIP_CU_SPLIT: finisher and scatter passes on disjoint CUs, two and three groups in flight,
# quarter workload; (3) the drop-in with the allocator tunables + registered pinned pool as defaults, 8 Gbp sweep and the 30 Gbp run
OUT=gpurun_out/r06o; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage1_e2e.py -m gpu -q -k "eight_logical or arena or redo_errors or several_bins_per_call or rank_path or writes_the_reference_database or reader_plugin or stage1" > $OUT/pytest_new.txt 2>&1; tail -4 $OUT/pytest_new.txt
bash tools/gpu_session.sh r06o qs:1:A=1 qs:2:A=1 qs:3:A=1 qs:2:KMC_HIP_CU_SPLIT=64 qs:2:KMC_HIP_CU_SPLIT=80 qs:2:KMC_HIP_CU_SPLIT=96 qs:3:KMC_HIP_CU_SPLIT=80 qs:3:KMC_HIP_CU_SPLIT=96 qs:4:KMC_HIP_CU_SPLIT=88 2>&1 | cut -c1-160
ENVS='[{}, {"KMC_HIP_TUNE_MALLOC": "0"}, {"KMC_HIP_TUNE_MALLOC": "0", "KMC_HIP_PINNED_POOL_MB": "1024"}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "4"}, {"KMC_HIP_READERS": "2"}, {"KMC_HIP_WORKER_GROUP": "1"}, {"KMC_HIP_PINNED_POOL_MB": "0"}]'
timeout 900 python tools/e2e_reader_sweep.py 8 "$ENVS" > $OUT/e2e_sweep_8gbp.jsonl 2> $OUT/e2e_sweep_8gbp.err; python - <<'PY'
import json
for ln in open("gpurun_out/r06o/e2e_sweep_8gbp.jsonl"):
    d=json.loads(ln); print(d["env"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("host_boundary") or "")[:230], "|", (d.get("timeline") or "")[88:330])
PY
timeout 1200 python tools/e2e_large_run.py 30 > $OUT/e2e_large_30gbp.json 2> $OUT/e2e_large_30gbp.err; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r06o/e2e_large_30gbp.json")); print({k:d[k] for k in d if k in ("ref_stage1_s","ref_stage2_s","hip_stage1_s","hip_stage2_s","speedup","hip_Gkmers_per_s","stats_equal","worker_report")}); print(d.get("timeline"))
except Exception as e: print("30gbp", e)
PY
