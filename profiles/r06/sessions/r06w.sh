#!/bin/bash
# round 6, session w: where the 30 Gbp drop-in's workers spend their engine seconds (host-boundary phase times), default allocator; pool size and reader threads
OUT=gpurun_out/r06w; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ENVS='[{}, {"KMC_HIP_PINNED_POOL_MB": "3072"}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "4", "KMC_HIP_PINNED_POOL_MB": "3072"}]'
timeout 900 python tools/e2e_reader_sweep.py 30 "$ENVS" > $OUT/e2e_sweep_30gbp.jsonl 2> $OUT/e2e_sweep_30gbp.err; python - <<'PY'
import json
for ln in open("gpurun_out/r06w/e2e_sweep_30gbp.jsonl"):
    d=json.loads(ln); print(d["env"], "rc", d["rc"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("host_boundary") or "")[:260], "|", (d.get("report") or "")[:250], "|", (d.get("timeline") or "")[88:330], (d.get("stderr_tail") or "")[-200:])
PY
free -g | head -2
