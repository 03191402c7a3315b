#!/bin/bash
# clang's static analyzer over the host side and the device side of every source of the library (--offload-host-only /
# --offload-device-only), reports from the ROCm headers left out.  No output = nothing found.   usage: bash tools/analyze_host.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/feabas_amd/csrc || exit 1
for side in host device; do
for f in fb_ctx fb_comm fb_match fb_ncc fb_ncc_ct fb_ncc_small fb_ncc_pfa fb_dog fb_solver fb_fem fb_pipeline fb_geom fb_render; do
  /opt/rocm/lib/llvm/bin/clang++ --analyze -x hip --offload-$side-only -std=c++17 --offload-arch=gfx950 -I../../include -I. -I/opt/rocm/include \
      -Wno-unused-value -Xclang -analyzer-output=text $f.hip 2>&1 | grep -E "warning:|error:" | grep -v "^/opt/rocm" | sed "s/^/$side $f: /"
done
done | sort -u
