#!/bin/bash
# round 6, session y: the final tree — smoke(), the whole -m gpu suite, bench.py as the driver runs it (detail kept), the 30 Gbp leg with more completer writers
OUT=gpurun_out/r06y; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt | cut -c1-200
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_detail.json $OUT/bench_detail.json 2>/dev/null; tail -c 1800 $OUT/bench.err
ENVS='[{}, {"KMC_HIP_WRITERS": "8"}, {"KMC_HIP_WRITERS": "8", "KMC_HIP_READERS": "4"}]'
timeout 900 python tools/e2e_reader_sweep.py 30 "$ENVS" > $OUT/e2e_sweep_30gbp.jsonl 2> $OUT/e2e_sweep_30gbp.err; python - <<'PY'
import json
for ln in open("gpurun_out/r06y/e2e_sweep_30gbp.jsonl"):
    d=json.loads(ln); print(d["env"], "rc", d["rc"], "s1", d["stage1_s"], "s2", d["stage2_s"], "wall", d["process_wall_s"], "|", (d.get("report") or "")[:120], "|", (d.get("timeline") or "")[88:330])
PY
free -g | head -2
