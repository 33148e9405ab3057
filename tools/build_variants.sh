#!/bin/bash
# Kernel A/B experiments: builds cpg_amd/lib/exp/libcpg_hip_<tag>.so with extra -D flags on one source.
# usage: tools/build_variants.sh conv3x3.hip tag1 "-DC3_EXP=1" tag2 "-DC3_EXP=2" ...   then CPG_HIP_LIB=<path> python tools/conv_bench.py
set -e
cd "$(dirname "$0")/.."
python -m cpg_amd.build >/dev/null
src=$1; shift
mkdir -p cpg_amd/lib/exp
others=$(ls cpg_amd/lib/*.o | grep -v "/${src%.*}.o")
while [ $# -gt 0 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function $flags -x hip -c cpg_amd/csrc/$src -o cpg_amd/lib/exp/${src%.*}_$tag.o &
done
wait
for o in cpg_amd/lib/exp/${src%.*}_*.o; do
  tag=$(basename $o .o); tag=${tag#${src%.*}_}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cpg_amd/lib/exp/libcpg_hip_$tag.so $o $others
done
ls -la cpg_amd/lib/exp/*.so
