#!/bin/bash
# A/B build of the engine with ONE translation unit replaced: tools/ab_build.sh NAME path/to/variant.hip [extra flags]
# -> libstacker_rs_amd/ab/libNAME.so (the other objects come from csrc/build). Use with STACKER_AMD_LIB=...
set -e
name=$1; src=$(realpath "$2"); shift 2              # (resolved before the cd below: a relative path is relative to the caller)
cd "$(dirname "$0")/../libstacker_rs_amd/csrc"
base=$(basename "$src")
orig=${ORIG:-$base}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -I. -I../../include"
case "$orig" in kernels_ecc_col.hip) FLAGS="$FLAGS -fno-slp-vectorize";; esac
mkdir -p ../ab /tmp/ab
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o /tmp/ab/$name.o "$src" 2>/dev/null
objs=$(ls build/*.o | grep -v "build/$orig.o")
/opt/rocm/bin/hipcc $FLAGS -shared -o ../ab/lib$name.so $objs /tmp/ab/$name.o -ldl -lpthread
echo built ../ab/lib$name.so
