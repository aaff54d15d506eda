#!/bin/bash
# round 6, session l: the -m gpu suite on the arena tree
OUT=gpurun_out/r06l; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt
