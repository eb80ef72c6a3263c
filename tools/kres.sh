#!/bin/bash
# tools/kres.sh FILE.hip ["-DFLAGS"]: static resources of every kernel of one source (VGPRs, scratch bytes, static LDS)
cd "$(dirname "$0")/../jxl-oxide_amd/csrc"
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function"
/opt/rocm/bin/hipcc $CXXFLAGS $2 -S --cuda-device-only -o /tmp/kres_$$.s $1 || exit 1
python3 - /tmp/kres_$$.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"  - \.agpr_count:.*?\.wavefront_size", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    print(f"{g('name')[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
rm -f /tmp/kres_$$.s
