#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAGS": libjxlgpu_NAME.so = the product objects with FILE.hip
# recompiled under extra flags (timing experiments; select with JXLGPU_LIB).
set -e
cd "$(dirname "$0")/../jxl-oxide_amd/csrc"
name=$1; file=$2; flags=$3
make -s -j8
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
/opt/rocm/bin/hipcc $CXXFLAGS $flags -c $file -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "\.prof\.o$" | grep -v "\.sched\.o$" | grep -v "^${file%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libjxlgpu_$name.so $objs /tmp/variant_$name.o
echo built libjxlgpu_$name.so
