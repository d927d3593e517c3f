#!/bin/bash
# Build container: the ISA of one kernel (default: the production render kernel) into /tmp/<name>.s, with its resource
# usage. usage: tools/debug/isa.sh [mangled-name-substring] [out.s]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
pat=${1:-render_items_kernelILb0ELb0ELb0}
out=${2:-/tmp/kernel.s}
make -C "$ROOT/pixel-art-raytracer_amd/csrc" asm 2>&1 | grep -A12 "Function Name: .*$pat" | grep -E "SGPRs:|VGPRs:|Spill|Occupancy" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//'
S="$ROOT/build/par_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
awk -v pat="$pat" '$0 ~ "^_Z.*" pat ".*:" {on=1} on {print} on && /\.end_amdhsa_kernel/ {exit}' "$S" > "$out"
echo "$out: $(wc -l < "$out") lines, $(grep -cE '^\s+v_' "$out") VALU, $(grep -cE '^\s+s_' "$out") SALU (static)"
