#!/bin/bash
# A/B builds of the library: recompiles the named sources with extra flags and links them with the objects of the
# regular build into ab/lib_<name>.so (select it at run time with FEABAS_HIP_LIB=ab/lib_<name>.so).
# usage: bash tools/build_variant.sh <name> "<extra flags>" fb_dog.hip [more.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/feabas_amd/csrc
make -C $CS -j8 > /dev/null
mkdir -p $ROOT/ab/obj_$NAME
OBJS=""
for f in $CS/build/*.o; do
  b=$(basename $f .o)
  case $b in test_*) continue;; esac          # (the objects of libfeabas_hip_test.so)
  use=$f
  for s in "$@"; do
    if [ "$b" == "$(basename $s .hip)" ]; then
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wno-unused-value -Wno-unused-result -fno-slp-vectorize $FLAGS -c $CS/$s -o $ROOT/ab/obj_$NAME/$b.o
      use=$ROOT/ab/obj_$NAME/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
hipcc --offload-arch=gfx950 $OBJS -shared -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib -o $ROOT/ab/lib_$NAME.so
echo built $ROOT/ab/lib_$NAME.so
