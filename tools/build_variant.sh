#!/bin/bash
# Fast diagnostic build: recompile ONE kernel file with -DCMF_DBG_<FLAG>... and link it against the production objects of the others.
#   tools/build_variant.sh conv_tangent_bf16x3 MFMAORD [MORE ...]  ->  cmf_amd/csrc/_obj/dbg_<FLAGS>.so
set -e
cd "$(dirname "$0")/.."
file=$1; shift
name=$(IFS=_; echo "$*")
defs=""; for f in "$@"; do defs="$defs -DCMF_DBG_$f"; done
python -m cmf_amd.build > /dev/null
extra=""; [ "$file" = conv_tangent_bf16x3 ] && extra="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=100000"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Icmf_amd/csrc $defs $extra -Wno-unused-command-line-argument \
  -c cmf_amd/csrc/$file.hip -o cmf_amd/csrc/_obj/dbg_${name}_$file.o
others=$(ls cmf_amd/csrc/_obj/*.o | grep -v "/dbg_" | grep -v "/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC cmf_amd/csrc/_obj/dbg_${name}_$file.o $others -o cmf_amd/csrc/_obj/dbg_${name}.so
echo built cmf_amd/csrc/_obj/dbg_${name}.so
