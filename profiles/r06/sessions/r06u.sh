#!/bin/bash
# round 6, session u: device memory of the stream slots reserved during stage 1 (kmc_hip_reserve_slot) + the reorder buffer: the drop-in tests, then the 8 Gbp sweep
# (default allocator: the shape of session n), slabs on / off
OUT=gpurun_out/r06u; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage1_e2e.py -m gpu -q -k "writes_the_reference_database or reader_plugin or stage1 or dropin or narrow_boundary or kff or several_bins_per_call or redo_errors" > $OUT/pytest_dropin.txt 2>&1; tail -3 $OUT/pytest_dropin.txt | cut -c1-200
ENVS='[{}, {"KMC_HIP_SLOT_SLAB_MB": "0"}, {}, {"KMC_HIP_SLOT_SLAB_MB": "0"}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "4"}, {"KMC_HIP_PINNED_POOL_MB": "2048"}]'
timeout 900 python tools/e2e_reader_sweep.py 8 "$ENVS" > $OUT/e2e_sweep_8gbp.jsonl 2> $OUT/e2e_sweep_8gbp.err; python - <<'PY'
import json
for ln in open("gpurun_out/r06u/e2e_sweep_8gbp.jsonl"):
    d=json.loads(ln); print(d["env"], "rc", d["rc"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("host_boundary") or "")[:260], "|", (d.get("report") or "")[:250], "|", (d.get("timeline") or "")[88:330], (d.get("stderr_tail") or "")[-200:])
PY
free -g | head -2
