#!/bin/bash
# A variant of the library with one translation unit rebuilt under extra flags:   bash tools/build_variant.sh <unit> <out.so> [-DFLAG ...]
# (on the GPU box or here; the other objects are the in-tree ones)
UNIT=$1; OUT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -w "$@" -I $ROOT/include -c $ROOT/nesvor_amd/csrc/$UNIT.hip -o $TMP/$UNIT.o || exit 1
OTHERS=$(ls $ROOT/nesvor_amd/lib/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $TMP/$UNIT.o $OTHERS -o $OUT
