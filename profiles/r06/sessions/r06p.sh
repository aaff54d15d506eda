#!/bin/bash
# round 6, session p (after r06o lost its box): the changed -m gpu cases that touch kernels only — no drop-in binaries, no RCCL, no CU masks
OUT=gpurun_out/r06p; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "arena or redo_errors or several_bins_per_call or rank_path" > $OUT/pytest_kernels.txt 2>&1; tail -4 $OUT/pytest_kernels.txt
