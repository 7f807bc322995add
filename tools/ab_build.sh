#!/bin/bash
# Developer A/B builds of libaa_interp.so: tools/ab_build.sh NAME "-DFLAG=1 ..." [file.hip ...]
# Recompiles the listed translation units (default: the headline u8 kernel's, aa_fused_u8_v3_c3.hip) with the extra flags into
# build_ab/NAME/ and links them with the stock objects into interpolate_antialiasing_amd/csrc/libaa_interp_NAME.so.
# Use with AA_INTERP_LIB=.../libaa_interp_NAME.so (see _lib.py).  Not part of the product build.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-aa_fused_u8_v3_c3.hip}
cd "$(dirname "$0")/../interpolate_antialiasing_amd/csrc"
mkdir -p build_ab/$NAME
OBJS=""
for f in aa_api aa_tables aa_generic aa_fused_u8 aa_fused_u8_v3 aa_fused_u8_v3_c1 aa_fused_u8_v3_c3 aa_fused_u8_v3_c4 aa_fused_u8_v3_c1f aa_fused_u8_v3_c3f aa_fused_u8_v3_c4f aa_fused_u8_v3_c1u aa_fused_u8_v3_c3u aa_fused_u8_v3_c4u aa_fused_float aa_fused_float_up aa_backward; do
  if echo " $FILES " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc $FLAGS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -c $f.hip -o build_ab/$NAME/$f.o &
    OBJS="$OBJS build_ab/$NAME/$f.o"
  else
    OBJS="$OBJS $f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libaa_interp_$NAME.so $OBJS
echo built libaa_interp_$NAME.so
