#!/bin/bash
# Builds libexa_hip variants with different -D... build-time constants (exa_device.h: EXA_MARCH_WAVES, EXA_KD_STACK, EXA_SEG_QUEUE ...) (here, no GPU needed) into build/variants/ (git-ignored,
# travels to the GPU box), and on the GPU box times each with bench.py on C4, twice, interleaved.
#   tools/ab_variants.sh build  name1:"-DEXA_SEG_QUEUE=5 ..." name2:"..."
#   tools/ab_variants.sh run [bench args]      -> gpurun_out/variants/results.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIBS="$ROOT/build/variants"
OUT="$ROOT/gpurun_out/variants"
CS="$ROOT/owlexabrick_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
KFLAGS=${KFLAGS--mllvm -amdgpu-sched-strategy=max-ilp}     # the Makefile's flags for the kernel translation units (KFLAGS= for none)
mode=$1; shift
if [ "$mode" = build ]; then
  mkdir -p "$OUT" "$LIBS"
  for spec in "$@"; do
    name=${spec%%:*}; defs=${spec#*:}
    (
      d="$OUT/obj_$name"; mkdir -p "$d"
      # the six kernel translation units (three forms x {stack walk and everything else, rope march}) side by side
      k() { /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $KFLAGS $defs "$@" -c "$CS/exa_kernels.hip"; }
      k -DEXA_BASIS_FORM=0 -o "$d/exa_kernels_f0.o" & k -DEXA_BASIS_FORM=1 -o "$d/exa_kernels_f1.o" &
      k -DEXA_BASIS_FORM=0 -DEXA_EMPTY_CELLS=1 -o "$d/exa_kernels_f0e.o" & k -DEXA_BASIS_FORM=0 -DEXA_TU_ROPE=1 -o "$d/exa_kernels_f0r.o" &
      k -DEXA_BASIS_FORM=1 -DEXA_TU_ROPE=1 -o "$d/exa_kernels_f1r.o" & k -DEXA_BASIS_FORM=0 -DEXA_EMPTY_CELLS=1 -DEXA_TU_ROPE=1 -o "$d/exa_kernels_f0er.o" &
      wait
      ls "$d"/exa_kernels_f0.o "$d"/exa_kernels_f1.o "$d"/exa_kernels_f0e.o "$d"/exa_kernels_f0r.o "$d"/exa_kernels_f1r.o "$d"/exa_kernels_f0er.o > /dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c "$CS/exa_lbvh.hip" -o "$d/exa_lbvh.o" &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -x hip -c "$CS/exa_module.cpp" -o "$d/exa_module.o" &&
      /opt/rocm/bin/hipcc $FLAGS -c "$CS/exa_prep.cpp" -o "$d/exa_prep.o" &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$LIBS/libexa_hip_$name.so" "$d"/*.o -lpthread &&
      rm -rf "$d" && echo "built $name ($defs)"
    ) 2>"$OUT/build_$name.err"
  done
  ls -la "$LIBS"/*.so
else
  mkdir -p "$OUT"
  : > "$OUT/results.txt"
  for rep in 1 2; do
    for so in "$LIBS"/libexa_hip_*.so; do
      name=$(basename "$so" .so); name=${name#libexa_hip_}
      EXA_HIP_LIB="$so" timeout -k 10 120 python3 "$ROOT/bench.py" --cpu-baseline off --pmc off --in-flight 1 --steps 20 --warmup 3 "$@" > "$OUT/$name.$rep.json" 2> "$OUT/$name.$rep.err" \
        && python3 -c "import json,sys; d=json.loads(open('$OUT/$name.$rep.json').read().strip().splitlines()[-1]); print('%-12s rep $rep  %.3f ms/frame  kernel %.3f ms  %.2f fps' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" | tee -a "$OUT/results.txt" \
        || { echo "$name failed" | tee -a "$OUT/results.txt"; tail -3 "$OUT/$name.$rep.err"; }
    done
  done
fi
