#!/bin/bash
# round 6, session m: the -m gpu suite on the arena tree (a278994 + tests), the bench line as the driver runs it, rocprofv3 + PMC of the full workload
bash tools/gpu_session.sh r06m facts tests bench prof pmc:FETCH_SIZE pmc:WRITE_SIZE 2>&1 | cut -c1-400
OUT=gpurun_out/r06m
python - <<'PY'
import json,glob
for f in ("gpurun_out/r06m/bench.json",):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(len(json.dumps(d)), json.dumps(d)[:3000])
    except Exception as e: print(f, e)
PY
for c in FETCH_SIZE WRITE_SIZE; do python tools/pmc_table.py $(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1) > $OUT/pmc_$c.txt 2>&1; done
f=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $f $w 189756000 $OUT/pmc_hbm_traffic.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
