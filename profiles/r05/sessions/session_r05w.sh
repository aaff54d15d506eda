#!/bin/bash
# round 5, session w: record widths outside BASELINE's (k = 32, 40, 64, 200; quarter workload): where the plan keeps the records in the passes (five or six top bytes)
OUT=gpurun_out/r05w; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest"
for k in 32 40 64; do
  timeout 600 python bench.py --k $k $Q > $OUT/k$k.json 2> $OUT/k$k.err
  python - <<PY
import json
d=json.loads(open("$OUT/k$k.json").read().strip().splitlines()[-1])
print("   k=$k: %.2f Gk-mers/s, %.1f ms/step (%d k-mers), %s, passes/record %.1f, moved %.0f B/k-mer, finisher %.3f ms, oracle %s, paths %s" % (d["value"], d["ms_per_step"], d["config"]["kmers"], d["roofline"]["kernel"], d["sort_path"]["hbm_passes_per_record"], d["sort_path"]["hbm_bytes_per_kmer_moved_by_design"], d["local_sort"]["avg_launch_ms"], [b["equal"] for b in d["self_check"]["oracle_bins"]], d["sort_path"]["groups_by_path"]))
PY
done
timeout 600 python bench.py --k 200 --leg custom --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest > $OUT/k200.json 2> $OUT/k200.err; tail -c 300 $OUT/k200.err
python - <<PY
import json
d=json.loads(open("$OUT/k200.json").read().strip().splitlines()[-1])
print("   k=200: %.2f Gk-mers/s, %.1f ms/step (%d k-mers), %s, passes/record %.1f, paths %s" % (d["value"], d["ms_per_step"], d["config"]["kmers"], d["roofline"]["kernel"], d["sort_path"]["hbm_passes_per_record"], d["sort_path"]["groups_by_path"]))
PY
