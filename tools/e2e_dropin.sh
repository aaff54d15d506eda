#!/bin/bash
# End-to-end drop-in check on the GPU box: reference kmc vs kmc_hip (reference pipeline + HIP stage-2 worker).
# usage: tools/e2e_dropin.sh <reads> <genome_len> [k]   -> prints stage times and md5 equality (-sr1 runs)
set -e
READS=${1:-1000000}; G=${2:-5000000}; K=${3:-27}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /dev/shm/kmce2e.XXXX)
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from kmc_amd import synth
print("fastq bytes", synth.make_fastq("$W/in.fq", seed=2026, genome_len=$G, n_reads=$READS))
PY
export KMC_HIP_LIB=${KMC_HIP_LIB:-$ROOT/kmc_amd/libkmc_hip.so}
for exe in kmc kmc_hip; do
  for mode in "-sr1" ""; do
    mkdir -p $W/tmp_$exe; rm -rf $W/tmp_$exe/*
    T0=$(date +%s.%N)
    $ROOT/oracle/_ref/$exe -k$K -m64 $mode $W/in.fq $W/out_${exe}${mode} $W/tmp_$exe > $W/log_${exe}${mode} 2>&1 || { tail -5 $W/log_${exe}${mode}; }
    T1=$(date +%s.%N)
    echo "== $exe $mode: $(grep -E '1st stage|2nd stage' $W/log_${exe}${mode} | tr '\n' ' ') wall $(python3 -c "print(round($T1 - $T0, 2))") s $(grep -E 'Total no. of k-mers|unique k-mers' $W/log_${exe}${mode} | tr -s ' ' | tr '\n' ' ')"
  done
done
md5sum $W/out_kmc-sr1.kmc_pre $W/out_kmc_hip-sr1.kmc_pre $W/out_kmc-sr1.kmc_suf $W/out_kmc_hip-sr1.kmc_suf | awk '{print $1}' | uniq -c
rm -rf $W
