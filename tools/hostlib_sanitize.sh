#!/bin/bash
# The PRODUCT'S HOST LIBRARY (kmc_amd/csrc/kmc_hip.hip over the emulated kernels and the emulated HIP runtime, tests/emu.py build_hostlib) under a
# sanitizer: out-of-bounds "device" accesses (asan: device memory is the heap there) or unsynchronised LDS / global accesses between the GPU threads of a
# workgroup (tsan: one OS thread per GPU thread). Usage: tools/hostlib_sanitize.sh asan|tsan. Builds into /tmp; runs tests/hostlib_sanitize_case.py.
set -u
kind=${1:-asan}
cd "$(dirname "$0")/.."
case $kind in
  asan) flag=address; lib=$(gcc -print-file-name=libasan.so); opts="ASAN_OPTIONS=detect_leaks=0" ;;
  tsan) flag=thread;  lib=$(gcc -print-file-name=libtsan.so); opts="TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=2" ;;
  *) echo "asan or tsan"; exit 2 ;;
esac
python -c "import sys; sys.path.insert(0, 'tests'); import emu; emu.build_hostlib('small')" || exit 1
flags=$(python -c "import sys; sys.path.insert(0, 'tests'); import emu; print(' '.join(emu.GEOMETRY_FLAGS['small']))")
so=/tmp/libkmc_hip_emu_$kind.so
g++ -std=c++17 -O1 -g -fPIC -shared -pthread -w -fno-gnu-unique -Wl,-Bsymbolic -DHIPEMU_HOST_API -fsanitize=$flag -fno-omit-frame-pointer $flags \
    -I tests/hipemu/include -I kmc_amd/csrc tests/hipemu/_gen/kmc_hip_emu_small.cpp -o $so || exit 1
log=/tmp/hostlib_$kind.log
env LD_PRELOAD=$lib $opts KMC_HIP_LIB=$so timeout 3400 python tests/hostlib_sanitize_case.py > $log 2>&1
echo "rc=$?; sanitizer reports: $(grep -c 'WARNING: ThreadSanitizer\|ERROR: AddressSanitizer' $log); log: $log"
tail -1 $log
