#!/bin/bash
# In-job A/B of the training step between builds of csrc/hashgrid.hip:  bash tools/ab_hashgrid_step.sh <out-dir> name:-DFLAG[,-DFLAG] ...
# (the other objects are taken from nesvor_amd/lib; tools/ab_step.sh alternates the libraries twice)
OUT=$1; shift
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p /tmp/abh $OUT
others=$(ls nesvor_amd/lib/*.o | grep -v hashgrid.o)
specs=""
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off $flags -I include -c nesvor_amd/csrc/hashgrid.hip -o /tmp/abh/$name.o 2>/dev/null &
  specs="$specs $name=/tmp/abh/lib$name.so"
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abh/$name.o $others -o /tmp/abh/lib$name.so
done
bash tools/ab_step.sh $OUT $specs
