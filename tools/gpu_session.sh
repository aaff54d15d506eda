#!/bin/bash
# One gpurun call = one measurement session; everything lands under gpurun_out/$TAG. Usage: tools/gpu_session.sh TAG step...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT /dev/shm/kmccache
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 KMC_SYNTH_VERBOSE=1
one_bin="--leg configs[1] --reads 13300000 --genome 66000000 --bins 1 --steps 3 --warmup 1 --no-digest"
bins512="--leg 2gbp-512bins --reads 13300000 --genome 66000000 --bins 512 --steps 3 --warmup 1 --no-digest"
for step in "$@"; do
  t0=$(date +%s)
  case $step in
    facts)   { nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null; free -g | head -2; df -h /dev/shm /tmp . | cat; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $OUT/facts.txt 2>&1 ;;
    tests)   timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;;
    bench)   timeout 1200 python bench.py --cache /dev/shm/kmccache > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.err ;;
    benchq)  timeout 900 python bench.py --no-cpu-baseline --no-secondary > $OUT/benchq.json 2> $OUT/benchq.err; tail -c 300 $OUT/benchq.err ;;
    e2e)     timeout 900 python tools/e2e_matrix.py > $OUT/e2e_matrix.jsonl 2> $OUT/e2e_matrix.err ;;
    e2es1)   timeout 600 python tools/e2e_matrix.py 13300000 66000000 27 2,3,15,16,17,18,19 > $OUT/e2e_stage1.jsonl 2> $OUT/e2e_stage1.err; python tools/pj.py $OUT/e2e_stage1.jsonl 2>/dev/null | cut -c1-400 ;;
    e2eq)    timeout 600 python tools/e2e_matrix.py 13300000 66000000 27 3,3,6 > $OUT/e2e_quick.jsonl 2> $OUT/e2e_quick.err ;;
    streams:*) n=${step#streams:}; timeout 900 python bench.py --cache /dev/shm/kmccache --streams $n --no-cpu-baseline --no-secondary --no-host-boundary --no-digest --steps 3 > $OUT/c3_streams$n.json 2> $OUT/c3_streams$n.err ;;
    small)   timeout 600 python bench.py --leg custom --reads 2000000 --genome 10000000 --bins 512 --steps 5 --warmup 1 --no-digest > $OUT/bins512small.json 2> $OUT/bins512small.err ;;
    small:*) n=${step#small:}; timeout 600 python bench.py --leg custom --reads 2000000 --genome 10000000 --bins 512 --steps 5 --warmup 1 --no-digest --streams $n > $OUT/bins512small_s$n.json 2> $OUT/bins512small_s$n.err; python tools/pj.py $OUT/bins512small_s$n.json | cut -c1-200 ;;
    b512)    timeout 600 python bench.py $bins512 > $OUT/bins512.json 2> $OUT/bins512.err ;;
    one)     timeout 600 python bench.py $one_bin > $OUT/onebin.json 2> $OUT/onebin.err ;;
    synth:*) n=${step#synth:}; KMC_SYNTH_VERBOSE=1 timeout 600 python -c "
import sys,time; sys.path.insert(0,'.')
from kmc_amd import capi
t=time.time(); s=capi.synth_bins(seed=2026, genome_len=1000000000, n_reads=200000000, k=27, n_bins=512, n_threads=$n, copy=False); print('threads $n total', time.time()-t)" > $OUT/synth_$n.txt 2>&1 ;;
    host)    timeout 600 python tools/ubench_host.py > $OUT/ubench_host.json 2> $OUT/ubench_host.err ;;
    var:*)   v=${step#var:}; KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_$v.so timeout 600 python bench.py $one_bin > $OUT/onebin_$v.json 2> $OUT/onebin_$v.err ;;
    var512:*) v=${step#var512:}; KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_$v.so timeout 600 python bench.py $bins512 > $OUT/bins512_$v.json 2> $OUT/bins512_$v.err ;;
    profk:*) a=${step#profk:}; k=${a%%:*}; e=${a#*:}; cd /tmp; env $e timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$OUT/profk_${k}_$e -o kt -- python $OLDPWD/bench.py --k $k --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest > $OLDPWD/$OUT/profk_${k}_$e.json 2> $OLDPWD/$OUT/profk_${k}_$e.err; cd $OLDPWD; find $OUT/profk_${k}_$e -name "*kernel_trace.csv" -delete; find $OUT/profk_${k}_$e -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-140 ;;
    pmck:*)  a=${step#pmck:}; k=${a%%:*}; b=${a#*:}; c=${b%%:*}; e=${b#*:}; cd /tmp; env $e timeout 900 rocprofv3 --pmc ${c//,/ } -f csv -d $OLDPWD/$OUT/pmck_${k}_${c}_$e -o pmc -- python $OLDPWD/bench.py --k $k --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 1 --warmup 0 --no-digest > $OLDPWD/$OUT/pmck_${k}_${c}_$e.json 2> $OLDPWD/$OUT/pmck_${k}_${c}_$e.err; cd $OLDPWD; python tools/pmc_table.py $OUT/pmck_${k}_${c}_$e/pmc_counter_collection.csv > $OUT/pmck_${k}_${c}_$e.txt 2>&1; rm -rf $OUT/pmck_${k}_${c}_$e; cat $OUT/pmck_${k}_${c}_$e.txt ;;
    qs:*)    a=${step#qs:}; n=${a%%:*}; e=${a#*:}; env $e timeout 600 python bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --streams $n > $OUT/qs_${n}_$e.json 2> $OUT/qs_${n}_$e.err; python tools/pj.py $OUT/qs_${n}_$e.json 2>&1 | cut -c1-120 ;;
    kq:*)    a=${step#kq:}; k=${a%%:*}; e=${a#*:}; env $e timeout 900 python bench.py --k $k --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest > $OUT/kq_${k}_$e.json 2> $OUT/kq_${k}_$e.err; python tools/pj.py $OUT/kq_${k}_$e.json 2>&1 | cut -c1-700 ;;
    kqv:*)   a=${step#kqv:}; k=${a%%:*}; v=${a#*:}; lib=kmc_amd/variants/libkmc_hip_$v.so; [ "$v" = base ] && lib=kmc_amd/libkmc_hip.so; KMC_HIP_LIB=$lib timeout 900 python bench.py --k $k --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest > $OUT/kqv_${k}_$v.json 2> $OUT/kqv_${k}_$v.err; python tools/pj.py $OUT/kqv_${k}_$v.json 2>&1 | cut -c1-330 ;;
    e2esample) bash tools/e2e_sample_main.sh $OUT/e2e_sample 2>&1 | cut -c1-900 ;;
    k:*)     k=${step#k:}; timeout 900 python bench.py --k $k --no-cpu-baseline --no-secondary --no-host-boundary --steps 3 > $OUT/bench_k$k.json 2> $OUT/bench_k$k.err ;;
    prof)    cd /tmp; timeout 1500 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$OUT/prof -o kt -- python $OLDPWD/bench.py --cache /dev/shm/kmccache --no-cpu-baseline --no-secondary --no-host-boundary --no-two-streams --no-digest --steps 3 > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err; cd $OLDPWD; find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-160 ;;
    pmc:*)   c=${step#pmc:}; cd /tmp; timeout 1500 rocprofv3 --pmc $c -f csv -d $OLDPWD/$OUT/pmc_$c -o pmc -- python $OLDPWD/bench.py --cache /dev/shm/kmccache --no-cpu-baseline --no-secondary --no-host-boundary --no-two-streams --no-digest --steps 1 --warmup 0 > $OLDPWD/$OUT/pmc_$c.json 2> $OLDPWD/$OUT/pmc_$c.err; cd $OLDPWD ;;
    q:*)     v=${step#q:}; lib=kmc_amd/variants/libkmc_hip_$v.so; [ "$v" = base ] && lib=kmc_amd/libkmc_hip.so; KMC_HIP_LIB=$lib timeout 600 python bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 > $OUT/q_$v.json 2> $OUT/q_$v.err; python tools/pj.py $OUT/q_$v.json 2>&1 | head -3 ;;
    profq:*) a=${step#profq:}; cd /tmp; env $a timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$OUT/profq_$a -o kt -- python $OLDPWD/bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 --no-digest > $OLDPWD/$OUT/profq_$a.json 2> $OLDPWD/$OUT/profq_$a.err; cd $OLDPWD; find $OUT/profq_$a -name "*kernel_trace.csv" -delete; find $OUT/profq_$a -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-160 ;;
    qe:*)    a=${step#qe:}; env $a timeout 600 python bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 3 --warmup 1 > $OUT/q_$a.json 2> $OUT/q_$a.err; python tools/pj.py $OUT/q_$a.json 2>&1 | head -3 ;;
    se:*)    a=${step#se:}; env $a timeout 600 python bench.py --leg custom --reads 2000000 --genome 10000000 --bins 512 --steps 5 --warmup 1 --no-digest > $OUT/small_$a.json 2> $OUT/small_$a.err; python tools/pj.py $OUT/small_$a.json | cut -c1-200 ;;
    tk:*)    e=${step#tk:}; timeout 900 python -m pytest tests -m gpu -x -q -k "$e" > $OUT/pytest_k.txt 2>&1; tail -3 $OUT/pytest_k.txt ;;
    pmcq:*)  a=${step#pmcq:}; v=${a%%:*}; c=${a#*:}; lib=kmc_amd/variants/libkmc_hip_$v.so; [ "$v" = base ] && lib=kmc_amd/libkmc_hip.so; cd /tmp; KMC_HIP_LIB=$OLDPWD/$lib timeout 600 rocprofv3 --pmc ${c//,/ } -f csv -d $OLDPWD/$OUT/pmcq_$v -o pmc -- python $OLDPWD/bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 1 --warmup 0 --no-digest > $OLDPWD/$OUT/pmcq_$v.json 2> $OLDPWD/$OUT/pmcq_$v.err; cd $OLDPWD; python tools/pmc_table.py $OUT/pmcq_$v/pmc_counter_collection.csv > $OUT/pmcq_$v.txt 2>&1; rm -rf $OUT/pmcq_$v; cat $OUT/pmcq_$v.txt ;;
    trace)   KMC_HIP_LIB=kmc_amd/variants/libkmc_hip_trace.so timeout 300 python tools/trace_run.py $TAG > $OUT/trace_run.txt 2>&1; python tools/trace_report.py gpurun_out/trace_$TAG.npy > $OUT/trace_report.txt 2>&1; rm -f gpurun_out/trace_$TAG.npy ;;
    s1prof)  cd /tmp; timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$OUT/s1prof -o kt -- python $OLDPWD/tools/s1_bench.py > $OLDPWD/$OUT/s1_bench.json 2> $OLDPWD/$OUT/s1_bench.err; cd $OLDPWD; cat $OUT/s1_bench.json; tail -2 $OUT/s1_bench.err; find $OUT/s1prof -name "*kernel_trace.csv" -delete ;;
    s1part)  timeout 120 python tools/s1_part_bench.py > $OUT/s1_part_bench.json 2> $OUT/s1_part_bench.err; cat $OUT/s1_part_bench.json; tail -2 $OUT/s1_part_bench.err ;;
    s1parts) KMC_HIP_S1_SORTED_EMIT=1 timeout 120 python tools/s1_part_bench.py > $OUT/s1_part_bench_sorted.json 2> $OUT/s1_part_bench_sorted.err; cat $OUT/s1_part_bench_sorted.json; tail -2 $OUT/s1_part_bench_sorted.err ;;
    b512e:*) e=${step#b512e:}; env $e timeout 600 python bench.py $bins512 --no-oracle-check > $OUT/bins512_$e.json 2> $OUT/bins512_$e.err; python tools/pj.py $OUT/bins512_$e.json | cut -c1-200 ;;
    onee:*)  e=${step#onee:}; env $e timeout 600 python bench.py $one_bin --no-oracle-check > $OUT/onebin_$e.json 2> $OUT/onebin_$e.err; python tools/pj.py $OUT/onebin_$e.json | cut -c1-200 ;;
    hbq:*)   e=${step#hbq:}; env $e timeout 600 python bench.py --cache /dev/shm/kmccache --reads 50000000 --genome 250000000 --bins 128 --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-two-streams --no-oracle-check --no-digest > $OUT/hbq_$e.json 2> $OUT/hbq_$e.err; python - <<PYEOF
import json
d=json.loads(open("$OUT/hbq_$e.json").read().strip().splitlines()[-1])
hb=d.get("host_boundary",{})
print("$e: value", round(d["value"],2), "host", d.get("value_host_boundary"), hb.get("error"), [(l["bins_per_call"], round(l["value"],2)) for l in hb.get("legs",[])])
PYEOF
    ;;
    hbprobe) timeout 900 python bench.py --cache /dev/shm/kmccache --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-oracle-check --no-digest --host-probe > $OUT/hbprobe.json 2> $OUT/hbprobe.err; python - <<PYEOF
import json
d=json.loads(open("$OUT/hbprobe.json").read().strip().splitlines()[-1])
print("value", round(d["value"],2))
for l in d.get("host_probe",[]): print(l)
PYEOF
    ;;
    pcie)    timeout 300 python tools/pcie_probe.py > $OUT/pcie_probe.json 2> $OUT/pcie_probe.err; cat $OUT/pcie_probe.json; tail -2 $OUT/pcie_probe.err ;;
    hb:*)    a=${step#hb:}; g=${a%%:*}; t=${a#*:}; timeout 600 python bench.py --cache /dev/shm/kmccache --leg quarter --reads 50000000 --genome 250000000 --bins 128 --steps 2 --warmup 1 --no-digest --no-oracle-check --no-two-streams --no-host-single --host-group $g --host-threads $t > $OUT/hb_${g}_$t.json 2> $OUT/hb_${g}_$t.err; python - <<PYEOF
import json
d=json.loads(open("$OUT/hb_${g}_$t.json").read().strip().splitlines()[-1])
print("group $g threads $t: value", round(d["value"],2), "host boundary", d.get("value_host_boundary"), d.get("host_boundary",{}).get("seconds"))
PYEOF
    ;;
    s1)      timeout 120 python tools/s1_bench.py > $OUT/s1_bench_plain.json 2> $OUT/s1_bench_plain.err; cat $OUT/s1_bench_plain.json; tail -2 $OUT/s1_bench_plain.err ;;
    *) echo "unknown step $step" ;;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s"
done
rm -rf /dev/shm/kmccache_never; find $OUT -name "*.db" -size +20M -delete 2>/dev/null
du -sh $OUT | cat
