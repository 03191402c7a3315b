#!/bin/bash
# The HOST code of the library under sanitizers, on the build machine (the GPU pool refuses sanitizer runs, and the device side
# cannot be instrumented without xnack): a second build of every source with -fsanitize=... (clang applies it to the host
# compilation only), then the CPU tests that call into the library and the host-entry fuzz against it.
#   bash tools/asan_host.sh [fuzz rounds]            AddressSanitizer + UBSan   (build in /tmp/feabas_asan)
#   SAN=thread bash tools/asan_host.sh [fuzz rounds] ThreadSanitizer            (build in /tmp/feabas_tsan; reports from inside
#                                                    numpy's OpenBLAS threads are not ours -- look for libfeabas frames)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=${SAN:-address,undefined}
if [ "$SAN" = thread ]; then OUT=/tmp/feabas_tsan; RTN=tsan; else OUT=/tmp/feabas_asan; RTN=asan; fi
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.$RTN-x86_64.so | head -1)
mkdir -p $OUT
cd $ROOT/feabas_amd/csrc || exit 1
for f in fb_ctx fb_comm fb_match fb_ncc fb_ncc_ct fb_ncc_small fb_ncc_pfa fb_dog fb_solver fb_fem fb_pipeline fb_geom fb_render; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ]; then
    ( hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value -Wno-unused-result -Wno-option-ignored \
        -fsanitize=$SAN -fno-omit-frame-pointer -DFB_TEST_HOOKS -c $f.hip -o $OUT/$f.o || echo "COMPILE FAILED $f" ) &
  fi
done
wait
hipcc --offload-arch=gfx950 $OUT/*.o -shared -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib -fsanitize=$SAN -shared-libsan -o $OUT/libfeabas_hip.so || exit 1
cd $ROOT
export FEABAS_HIP_LIB=$OUT/libfeabas_hip.so FEABAS_HIP_TEST_LIB=$OUT/libfeabas_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
       TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0
python -m pytest tests/test_cpu_host.py tests/test_cpu_region_distributor.py tests/test_cpu_matcher_loop.py -q -p no:cacheprovider 2>&1 | tail -3 &&
python tools/fuzz_host_entries.py 1 ${1:-60} 2>&1 | grep -v "^$" | tail -40
