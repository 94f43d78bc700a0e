#!/bin/bash
# usage: tools/build_variant.sh <name> <-D flags ...>  ->  quda-qkxtm-multigrid_amd/lib/libquda_<name>.so: the library with the stencil file
# recompiled under the given macros (A/B timing of cache policies: QUDA_AMD_LIBRARY=<that file> python bench.py ...)
set -e
name=$1; shift
cd "$(dirname "$0")/../quda-qkxtm-multigrid_amd"
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -Icsrc -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=fast \
  -fno-slp-vectorize "$@" -c csrc/dslash.hip -o build_var/dslash_$name.o
objs=$(ls build/*.o | grep -v "build/dslash.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o lib/libquda_$name.so $objs build_var/dslash_$name.o
echo built lib/libquda_$name.so
