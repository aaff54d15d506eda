#!/bin/bash
# round 6, session t: the whole -m gpu suite on the final tree, bench.py --logical 8 on a small bin set, then bench.py as the driver runs it (its detail record kept)
OUT=gpurun_out/r06t; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt | cut -c1-200
timeout 300 python bench.py --reads 2000000 --genome 10000000 --bins 64 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-host-boundary --logical 8 > $OUT/bench_small_logical8.json 2> $OUT/bench_small_logical8.err; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r06t/bench_small_logical8.json").read().strip().splitlines()[-1]); print("small:", d["value"], d.get("logical_devices"))
except Exception as e: print("small logical", e)
PY
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_detail.json $OUT/bench_detail.json 2>/dev/null; tail -c 1500 $OUT/bench.err
free -g | head -2
