#!/bin/bash
# usage: tools/kernel_regs.sh file.o [name-filter]   -> kernel name, vgpr_count, sgpr_count, spills of a hipcc object's gfx950 code
set -e
OBJ=$(realpath "$1"); T=$(mktemp -d); cp "$OBJ" $T/o.o; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null
CO=$(ls | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | awk '/\.name:/{n=$2} /\.sgpr_count:/{s=$2} /\.sgpr_spill_count:/{ss=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "spill", ss "/" $2}' | grep -- "${2:-.}" || true
rm -rf $T
