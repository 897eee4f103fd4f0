#!/bin/bash
# The lab build of the library (-DORBFE_EXPERIMENTS: measurement switches, instrumented kernels) beside the product one:
# os1_amd/liborbfe_exp.so, objects under build/exp.  Use it with ORBFE_LIB=os1_amd/liborbfe_exp.so.
set -e
cd "$(dirname "$0")/../os1_amd/csrc"
mkdir -p ../../build/exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -DORBFE_EXPERIMENTS"
objs=""
for f in orbfe_kernels orbfe_fast orbfe_quadtree orbfe_sfi orbfe_extractor orbfe_matcher orbfe_frame orbfe_bow; do
  /opt/rocm/bin/hipcc $FLAGS -c -o ../../build/exp/$f.o $f.hip &
  objs="$objs ../../build/exp/$f.o"
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -DORBFE_EXPERIMENTS -c -o ../../build/exp/orbfe_stream.o orbfe_stream.cpp &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -DORBFE_EXPERIMENTS -c -o ../../build/exp/orbfe_stream_multi.o orbfe_stream_multi.cpp &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbfe_exp.so $objs ../../build/exp/orbfe_stream.o ../../build/exp/orbfe_stream_multi.o -lpthread
echo built os1_amd/liborbfe_exp.so
