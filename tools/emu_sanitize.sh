#!/bin/bash
# Kernel source under the CPU emulation + a sanitizer: out-of-bounds accesses (asan) or data races between GPU threads (tsan) in
# kmc_amd/csrc/kernels.hip.h / stage1_kernels.hip.h, found on a box without a GPU (ubsan: shifts, overflows, misaligned typed accesses).
# Usage: tools/emu_sanitize.sh asan|tsan|ubsan [pytest -k expression]
# Builds a sanitized copy of tests/hipemu/libkmc_emu_small.so, runs the emulated kernel tests against it, restores the plain library.
# A clean run prints only pytest's summary; every sanitizer report is in the log it names.
set -u
kind=${1:-asan}; expr=${2:-"not product"}
cd "$(dirname "$0")/.."
case $kind in
  asan) flag=address; lib=$(gcc -print-file-name=libasan.so); opts="ASAN_OPTIONS=detect_leaks=0" ;;
  tsan) flag=thread;  lib=$(gcc -print-file-name=libtsan.so); opts="TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=2" ;;
  ubsan) flag=undefined; lib=$(gcc -print-file-name=libubsan.so); opts="UBSAN_OPTIONS=print_stacktrace=0" ;;
  *) echo "asan, tsan or ubsan"; exit 2 ;;
esac
python -c "import sys; sys.path.insert(0, 'tests'); import emu; emu.build('small')" || exit 1
so=tests/hipemu/libkmc_emu_small.so
flags=$(python -c "import sys; sys.path.insert(0, 'tests'); import emu; print(' '.join(emu.GEOMETRY_FLAGS['small']))")
cp $so /tmp/libkmc_emu_small.plain
g++ -std=c++17 -O1 -g -fPIC -shared -pthread -w -fno-gnu-unique -fvisibility=hidden -Wl,-Bsymbolic -fsanitize=$flag -fno-omit-frame-pointer $flags \
    -I tests/hipemu/include tests/hipemu/emu_kernels.cpp -o $so || { cp /tmp/libkmc_emu_small.plain $so; exit 1; }
touch $so
log=/tmp/emu_${kind}.log
env LD_PRELOAD=$lib $opts timeout 3400 python -m pytest tests/test_stage1_emulated.py tests/test_kernels_emulated.py -q -s -p no:cacheprovider -k "$expr" > $log 2>&1
rc=$?
cp /tmp/libkmc_emu_small.plain $so; touch $so
echo "pytest rc=$rc; sanitizer reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:' $log); log: $log"
tail -2 $log
