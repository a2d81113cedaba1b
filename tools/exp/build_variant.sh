#!/bin/bash
# experiment builds of the library: spsp_compare.hip compiled with -DSPSP_EXP=<bits> (parts of a kernel's work left out --
# results are WRONG, the point is what the rest costs), linked with the product's other objects into tools/exp/ab/libspsp_exp<bits>.so
# (select with SPSP_LIB).  usage (in the build container): tools/exp/build_variant.sh <bits> [<bits> ...]
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/supersampler_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $R/tools/exp/ab
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DSPSP_EXP=$v -c $C/spsp_compare.hip -o /tmp/spsp_compare_exp$v.o
  objs=$(ls $C/*.o | grep -v spsp_compare.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/tools/exp/ab/libspsp_exp$v.so /tmp/spsp_compare_exp$v.o $objs -lz -lpthread
  echo built tools/exp/ab/libspsp_exp$v.so
done
