#!/bin/bash
# tools/build_variant.sh NAME "-DFLAGS" FILE.hip [FILE2.hip ...]: libjxlgpu_NAME.so = the product objects with the named
# sources recompiled under extra flags (timing experiments; select with JXLGPU_LIB).
set -e
cd "$(dirname "$0")/../jxl-oxide_amd/csrc"
name=$1; flags=$2; shift 2
make -s -j8
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
objs=$(ls *.o | grep -v "\.prof\.o$" | grep -v "\.sched\.o$")
vobjs=""
for file in "$@"; do
  /opt/rocm/bin/hipcc $CXXFLAGS $flags -c $file -o /tmp/variant_${name}_${file%.hip}.o &
  objs=$(echo "$objs" | grep -v "^${file%.hip}.o$")
  vobjs="$vobjs /tmp/variant_${name}_${file%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libjxlgpu_$name.so $objs $vobjs
echo built libjxlgpu_$name.so
