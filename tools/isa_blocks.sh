#!/bin/bash
# tools/isa_blocks.sh FILE.hip KERNEL_MANGLED_NAME ["-DFLAGS"]: per-basic-block VALU / packed / DPP / mov counts
# of one kernel (static view of a loop body), plus its register counts.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/jxl-oxide_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I../../include $3 --cuda-device-only -S $1 -o /tmp/isa_blocks.s 2>&1 | grep -E "error" -A5
python3 $R/tools/isa_blocks.py /tmp/isa_blocks.s $2
grep -E "\.name: +$2" -A12 /tmp/isa_blocks.s | grep -E "vgpr_count|sgpr_count|private_seg"
