#!/bin/bash
# Registers / scratch / LDS of the kernels in csrc/exa_kernels_f0.o (EXA_FORM=1: _f1.o, EXA_FORM=1r: the rope march of form 1, ...;
# gfx950 code object), optionally filtered by a regex.
# usage: tools/kernel_resources.sh [regex]   (run after `make -C owlexabrick_amd/csrc`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
cd "$TMP"
cp "$ROOT/owlexabrick_amd/csrc/exa_kernels_f${EXA_FORM:-0}.o" k.o          # EXA_FORM=1: the per-axis association
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o > /dev/null
CO=$(ls | grep gfx950)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|\.group_segment_fixed_size|vgpr_spill" \
  | paste - - - - - - | sed 's/ \+/ /g; s/\t/ /g' | grep -E "${1:-.}" \
  | awk '{for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1); if($i==".vgpr_count:")v=$(i+1); if($i==".sgpr_count:")s=$(i+1); if($i==".private_segment_fixed_size:")p=$(i+1); if($i==".vgpr_spill_count:")sp=$(i+1); if($i==".group_segment_fixed_size:")l=$(i+1)} printf "vgpr %3d sgpr %3d scratch %4d spill %3d lds %5d  %s\n", v,s,p,sp,l,n}'
cp "$CO" /tmp/isa/k.co 2>/dev/null || true
rm -rf "$TMP"
