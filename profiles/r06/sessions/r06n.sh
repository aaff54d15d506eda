#!/bin/bash
# round 6, session n: (1) the drop-in's host side at 8 Gbp under allocator / reader settings, with the host-boundary phase times; (2) how the kernels scale with the CUs they
# may use (HSA_CU_MASK on the quarter workload under rocprofv3: is k_onesweep bound by the memory system or by the CUs?); (3) the new 8-logical-device test
OUT=gpurun_out/r06n; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eight_logical or arena or redo_errors or several_bins_per_call or rank_path" > $OUT/pytest_new.txt 2>&1; tail -3 $OUT/pytest_new.txt
ENVS='[{}, {"GLIBC_TUNABLES": "glibc.malloc.hugetlb=1"}, {"MALLOC_MMAP_MAX_": "0", "MALLOC_TRIM_THRESHOLD_": "1099511627776", "MALLOC_TOP_PAD_": "1073741824"}, {"KMC_HIP_READERS": "16"}, {"KMC_HIP_READERS": "4"}, {"KMC_HIP_PINNED_POOL_MB": "8192"}, {"KMC_HIP_WORKER_GROUP": "1"}, {"GLIBC_TUNABLES": "glibc.malloc.hugetlb=1", "KMC_HIP_PINNED_POOL_MB": "8192", "KMC_HIP_READERS": "16"}]'
timeout 900 python tools/e2e_reader_sweep.py 8 "$ENVS" > $OUT/e2e_sweep_8gbp.jsonl 2> $OUT/e2e_sweep_8gbp.err; cut -c1-1200 $OUT/e2e_sweep_8gbp.jsonl
bash tools/gpu_session.sh r06n "profk:27:HSA_CU_MASK=0:0-191" "profk:27:HSA_CU_MASK=0:0-127" "profk:27:ROC_GLOBAL_CU_MASK=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF" 2>&1 | cut -c1-200
