bash tools/pmc_pw.sh
echo "== default"; python tools/generic_bench.py --iters 10 --only "1x1" 2>&1 | grep -v amdgpu
echo "== minw2"; CPG_HIP_LIB=$PWD/cpg_amd/lib/exp/libcpg_hip_minw2.so python tools/generic_bench.py --iters 10 --only "1x1" 2>&1 | grep -v amdgpu
