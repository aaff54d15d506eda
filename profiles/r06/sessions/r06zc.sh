#!/bin/bash
# round 6, session zc: large blocks freed with delete[] (the reference's bin parts in RAM-only mode) handed to a background thread instead of being unmapped inside the
# reader's critical path: the drop-in tests, then A/B at 8 and 30 Gbp (default allocator)
OUT=gpurun_out/r06zc; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage1_e2e.py -m gpu -q -k "writes_the_reference_database or reader_plugin or stage1 or dropin or narrow_boundary or kff" > $OUT/pytest_dropin.txt 2>&1; tail -2 $OUT/pytest_dropin.txt | cut -c1-200
show() { python - "$1" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    d=json.loads(ln); print(d["env"], "rc", d["rc"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("host_boundary") or "")[:200], "|", (d.get("report") or "")[:130], "|", (d.get("timeline") or "")[88:330], (d.get("stderr_tail") or "")[-200:])
PY
}
ENVS='[{}, {"KMC_HIP_DEFER_FREE_MB": "0"}, {}, {"KMC_HIP_DEFER_FREE_MB": "0"}]'
timeout 900 python tools/e2e_reader_sweep.py 8 "$ENVS" > $OUT/e2e_sweep_8gbp.jsonl 2> $OUT/e2e_sweep_8gbp.err; show $OUT/e2e_sweep_8gbp.jsonl
free -g | head -2
ENVS='[{}, {"KMC_HIP_DEFER_FREE_MB": "0"}, {}]'
timeout 900 python tools/e2e_reader_sweep.py 30 "$ENVS" > $OUT/e2e_sweep_30gbp.jsonl 2> $OUT/e2e_sweep_30gbp.err; show $OUT/e2e_sweep_30gbp.jsonl
free -g | head -2
