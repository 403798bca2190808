cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05y
L=nesvor_amd/lib/libnesvor_hip.so
bash tools/ab_step.sh gpurun_out/r05y s1=$L,NESVOR_OWNER_STRIDE=1 s397=$L,NESVOR_OWNER_STRIDE=397 s33=$L,NESVOR_OWNER_STRIDE=33 s129=$L,NESVOR_OWNER_STRIDE=129 s7=$L,NESVOR_OWNER_STRIDE=7 2>&1 | cut -c1-200
