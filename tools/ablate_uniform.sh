set -e
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/v gpurun_out/r05u
others=$(ls nesvor_amd/lib/*.o | grep -v hashgrid.o)
for v in 0 8 16 24; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DNESVOR_ABLATE=$v -I include -c nesvor_amd/csrc/hashgrid.hip -o /tmp/v/hg$v.o 2>/dev/null &
done
wait
for v in 0 8 16 24; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/v/hg$v.o $others -o /tmp/v/lib$v.so
  echo "== ablate $v (8: no record writes, 16: no queue reservations)"
  NESVOR_HIP_LIB=/tmp/v/lib$v.so python tools/bench_hg_levels.py U 0 2>&1 | grep -v amdgpu | tail -6
  echo "-- no input gradient"
done > gpurun_out/r05u/ablate_uniform.log 2>&1
cat gpurun_out/r05u/ablate_uniform.log
