#!/bin/bash
# round 6, session r: the 8-logical-device case on its own (RCCL over duplicate devices), bounded
OUT=gpurun_out/r06r; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 180 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "eight_logical" > $OUT/pytest_logical.txt 2>&1; echo "rc $?"; grep -v "^$" $OUT/pytest_logical.txt | tail -12 | cut -c1-300
