#!/bin/bash
# Diagnostic builds of libcmf_amd.so: tools/build_dbg.sh STAMP [NOMFMA ...] -> cmf_amd/csrc/_obj/dbg_<FLAGS>.so
# (every flag X becomes -DCMF_DBG_X; timing-only variants compute wrong results by design)
set -e
cd "$(dirname "$0")/.."
name=$(IFS=_; echo "$*")
defs=""; for f in "$@"; do defs="$defs -DCMF_DBG_$f"; done
mkdir -p cmf_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Icmf_amd/csrc $defs \
  -fno-slp-vectorize -Wno-unused-command-line-argument cmf_amd/csrc/*.hip -o cmf_amd/csrc/_obj/dbg_${name}.so
echo built cmf_amd/csrc/_obj/dbg_${name}.so
