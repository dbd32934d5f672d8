#!/bin/bash
# sha256 (first 16 hex digits) and size of the gfx950 .text of each kernel translation unit (after `make -C owlexabrick_amd/csrc`):
# a refactoring that is meant to leave the machine code alone shows the same three hashes before and after.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for f in "$ROOT"/owlexabrick_amd/csrc/exa_kernels_*.o; do
  TMP=$(mktemp -d)
  ( cd "$TMP" && cp "$f" k.o && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading k.o > /dev/null \
    && /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.text "$(ls | grep gfx950)" text.bin \
    && printf "%s  %s bytes  %s\n" "$(sha256sum text.bin | cut -c1-16)" "$(stat -c %s text.bin)" "$(basename "$f")" )
  rm -rf "$TMP"
done
