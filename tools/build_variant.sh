#!/bin/bash
# Fast diagnostic build: recompile ONE kernel file with -DCMF_DBG_<FLAG>... and link it against the production objects of the others.
#   tools/build_variant.sh conv_tangent_bf16x3 MFMAORD [MORE ...]  ->  cmf_amd/csrc/_obj/dbg_<FLAGS>.so
set -e
cd "$(dirname "$0")/.."
file=$1; shift
name=$(IFS=_; echo "$*")
defs=""; for f in "$@"; do defs="$defs -DCMF_DBG_$f"; done
python -m cmf_amd.build > /dev/null
# the product build's per-file flags (cmf_amd/build.py PER_FILE_FLAGS): a variant must differ from the product only by its -D switches
extra=$(python -c "from cmf_amd import build as B; print(' '.join(B.PER_FILE_FLAGS.get('$file.hip', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Icmf_amd/csrc $defs $extra -Wno-unused-command-line-argument \
  -c cmf_amd/csrc/$file.hip -o cmf_amd/csrc/_obj/dbg_${name}_$file.o
others=$(ls cmf_amd/csrc/_obj/*.o | grep -v "/dbg_" | grep -v "/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC cmf_amd/csrc/_obj/dbg_${name}_$file.o $others -o cmf_amd/csrc/_obj/dbg_${name}.so
echo built cmf_amd/csrc/_obj/dbg_${name}.so
