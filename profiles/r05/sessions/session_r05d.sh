#!/bin/bash
# round 5, session d: the bucket-size spectrum of the three quarter workloads (uniform / one repeat family / a spectrum of families)
OUT=gpurun_out/r05d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/bucket_hist.py --bins-sampled 6 > $OUT/bucket_hist_uniform.txt 2> $OUT/bh_u.err; cat $OUT/bucket_hist_uniform.txt
KMC_SYNTH_REPEATS=10000:2000:10 timeout 300 python tools/bucket_hist.py --bins-sampled 6 > $OUT/bucket_hist_skew.txt 2> $OUT/bh_s.err; cat $OUT/bucket_hist_skew.txt
KMC_SYNTH_REPEATS=300:100000:120,6000:5000:20,171:100000:20,H20000 timeout 300 python tools/bucket_hist.py --bins-sampled 6 > $OUT/bucket_hist_spectrum.txt 2> $OUT/bh_p.err; cat $OUT/bucket_hist_spectrum.txt
tail -3 $OUT/bh_*.err
