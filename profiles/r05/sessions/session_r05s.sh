#!/bin/bash
OUT=gpurun_out/r05s; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/e2e_reader_sweep.py 30 2>&1 | tee $OUT/reader_sweep.jsonl | cut -c1-700
