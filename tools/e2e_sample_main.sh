#!/bin/bash
# Where does the main thread of the drop-in spend the time after the completer has closed its files? (round-3 verdict item 7)
# usage (on the GPU box): tools/e2e_sample_main.sh OUTDIR — the 2 Gbp FASTQ through kmc_amd/bin/kmc_hip with KMC_HIP_SAMPLE_MAIN=2 (a sample every 2 ms)
OUT=${1:-gpurun_out/e2e_sample}; mkdir -p $OUT /dev/shm/e2es
python - <<PY
import sys; sys.path.insert(0, ".")
from kmc_amd import capi
capi.synth_fastq("/dev/shm/e2es/s.fq", seed=2026, genome_len=66000000, n_reads=13300000)
PY
for i in 1 2; do
  KMC_HIP_VERBOSE=1 KMC_HIP_SAMPLE_MAIN=2 kmc_amd/bin/kmc_hip -k27 -t128 -m128 -sr16 -hp /dev/shm/e2es/s.fq /dev/shm/e2es/db /dev/shm/e2es > $OUT/run$i.out 2> $OUT/run$i.err
  grep "2nd stage\|1st stage" $OUT/run$i.out
  grep "timeline" $OUT/run$i.err
done
# the samples of the last 0.4 s before exit, frames only
grep -A100000 "main-thread samples" $OUT/run2.err | tail -200 | cut -c1-260 > $OUT/run2_last_samples.txt
rm -rf /dev/shm/e2es
