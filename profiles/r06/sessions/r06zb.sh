#!/bin/bash
# round 6, session zb: the FINAL tree — smoke(), the whole -m gpu suite, bench.py as the driver runs it; then rocprofv3 kernel statistics of the quarter legs (uniform, one
# repeat family, spectrum, k = 55, k = 127) and the SQ instruction counters of k = 27 (uniform, spectrum), each in its own run
OUT=gpurun_out/r06zb; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt | cut -c1-200
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cp bench_detail.json $OUT/bench_detail.json 2>/dev/null; grep "^\[bench\] configs" $OUT/bench.err | cut -c1-220
SPEC=300:100000:120,6000:5000:20,171:100000:20,H20000
bash tools/gpu_session.sh r06zb profk:27:A=1 profk:27:KMC_SYNTH_REPEATS=10000:2000:10 profk:27:KMC_SYNTH_REPEATS=$SPEC profk:55:A=1 profk:127:A=1 pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:A=1 pmck:27:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES:KMC_SYNTH_REPEATS=$SPEC > $OUT/gpu_session.log 2>&1; grep "^\[" $OUT/gpu_session.log | cut -c1-120
mv "$OUT/profk_27_A=1" $OUT/p_uniform; mv "$OUT/profk_27_KMC_SYNTH_REPEATS=10000:2000:10" $OUT/p_skew; mv "$OUT/profk_27_KMC_SYNTH_REPEATS=$SPEC" $OUT/p_spectrum; mv "$OUT/profk_55_A=1" $OUT/p_k55; mv "$OUT/profk_127_A=1" $OUT/p_k127
mv "$OUT/pmck_27_SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES_A=1.txt" $OUT/pmc_instructions_k27.txt; mv "$OUT/pmck_27_SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVES_KMC_SYNTH_REPEATS=$SPEC.txt" $OUT/pmc_instructions_k27_spectrum.txt
ls $OUT
