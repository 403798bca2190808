#!/bin/bash
# A/B builds of csrc/hashgrid.hip timed on uniform points through the unclustered backward (tools/prof_hg_uniform.py):
#   bash tools/variants_uniform.sh name:-DFLAG[,-DFLAG][@ENV=VAL] ...
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p /tmp/vu
others=$(ls nesvor_amd/lib/*.o | grep -v hashgrid.o)
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; flags=${rest%%@*}; flags=${flags//,/ }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off $flags -I include -c nesvor_amd/csrc/hashgrid.hip -o /tmp/vu/$name.o 2>/dev/null &
done
wait
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=""
  if [[ "$rest" == *@* ]]; then envs=${rest#*@}; fi
  [ -f /tmp/vu/lib$name.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/vu/$name.o $others -o /tmp/vu/lib$name.so
  echo -n "$name ($rest) round $round: "
  env NESVOR_HIP_LIB=/tmp/vu/lib$name.so $envs python tools/prof_hg_uniform.py 1 2>&1 | grep uniform
done
done
