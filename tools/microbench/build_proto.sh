#!/bin/bash
# builds proto_mfma and its ablation / ring-depth variants next to the source (developer tool)
cd "$(dirname "$0")"
INC=../../interpolate_antialiasing_amd/csrc
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -I$INC"
$HC -o proto_mfma proto_mfma.hip &
for a in 1 2 3 4 5 6 7; do $HC -DAA_MFMA_ABL=$a -o proto_mfma_abl$a proto_mfma.hip & done
for r in "$@"; do $HC -DPROTO_R=$r -o proto_mfma_r$r proto_mfma.hip & done
wait
