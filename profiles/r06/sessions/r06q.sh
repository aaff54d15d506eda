#!/bin/bash
# round 6, session q: the -m gpu cases that run the drop-in binaries (allocator tunables by re-exec, registered pinned pool)
OUT=gpurun_out/r06q; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stage1_e2e.py -m gpu -q -k "writes_the_reference_database or reader_plugin or stage1 or dropin or narrow_boundary or kff" > $OUT/pytest_dropin.txt 2>&1; tail -4 $OUT/pytest_dropin.txt
free -g | head -2
