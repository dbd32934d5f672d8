#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code, driven by the seeded random inputs of tests/fuzz_*.py
# (no GPU: device ASan is not available on this pool).  Builds into a scratch directory:
#   exaBuilder (owlexabrick_amd/host/exa_builder.cpp)          <- tests/fuzz_builder.py, 300 cell sets
#   the oracle (oracle/exa_oracle.c, + float-cast-overflow)    <- tests/fuzz_oracle.py, fuzz_spec.py, whole frames of 3 case families
#   the module's host preparation (csrc/exa_prep.cpp) and the rope construction (csrc/exa_ropes.h)
#                                                               <- tests/fuzz_prep.py, 600 scenes + 200 (needs csrc/*.o: run make first)
#   the facade's file loaders (host/exa_host.cpp via exaRender --info) <- 150 random scenes written in the reference's formats
# usage: tools/sanitize_host.sh [scratch dir]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=${1:-/tmp/exa_san}
mkdir -p "$D"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -O1 -g"
g++ $SAN -std=c++17 -o "$D/exaBuilder" "$ROOT/owlexabrick_amd/host/exa_builder.cpp" -lpthread
gcc $SAN -fsanitize=float-cast-overflow -fno-sanitize-recover=float-cast-overflow -std=c11 -fPIC -ffp-contract=off -fno-fast-math -D_GNU_SOURCE \
    -shared -o "$D/libexa_oracle.so" "$ROOT/oracle/exa_oracle.c" -lm -lpthread
CS="$ROOT/owlexabrick_amd/csrc"
/opt/rocm/bin/hipcc $SAN -std=c++17 -fPIC -ffp-contract=off -c "$CS/exa_prep.cpp" -o "$D/exa_prep.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$D/libexa_hip.so" \
    "$CS/exa_kernels_f0.o" "$CS/exa_kernels_f1.o" "$CS/exa_kernels_f0e.o" "$CS/exa_kernels_f0r.o" "$CS/exa_kernels_f1r.o" "$CS/exa_kernels_f0er.o" \
    "$CS/exa_lbvh.o" "$CS/exa_module.o" "$D/exa_prep.o" -lpthread 2>/dev/null
GCC_RT="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
CLANG_RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
export ASAN_OPTIONS=detect_leaks=0
cd "$ROOT/tests"
python3 - "$D" <<'PY'
import sys
D = sys.argv[1]
sys.path.insert(0, '.')
import test_builder
test_builder.EXE = D + "/exaBuilder"
import fuzz_builder
fails = sum(bool(fuzz_builder.check(s)[0]) for s in range(300))
# ... and with --allow-empty-cells on cell sets with a quarter of the cells removed
import numpy as np
from owlexabrick_amd import scenes
from oracle import builder_oracle as bo
for s in range(40):
    rng = np.random.default_rng(0x5A17 + s)
    full = test_builder.cells_of(scenes.amr(seed=3 + s, root=(2, 2, 1), B=4, levels=2))
    cells = full[rng.uniform(size=len(full)) > 0.25]
    r, data = test_builder.run_builder(cells, ["--allow-empty-cells"], max_leaf_width=6)
    fails += int(r.returncode != 0 or data != bo.to_bricks_file_bytes(bo.build_bricks(cells, bo.SAH_ALIKE, max_leaf_width=6, allow_empty_cells=True)))
print(f"exaBuilder under ASan+UBSan: {fails} failed of 340")
sys.exit(1 if fails else 0)
PY
LD_PRELOAD="$GCC_RT" python3 - "$D" <<'PY'
import sys
D = sys.argv[1]
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from oracle import pyoracle
pyoracle._LIB = D + "/libexa_oracle.so"
import fuzz_oracle, fuzz_spec
from fuzz_cases import random_case, random_rich_case
fails = sum(bool(m.check(s)[0]) for s in range(120) for m in (fuzz_oracle, fuzz_spec))
for s in range(60):
    c, d, f = random_rich_case(s); c.run_oracle(frames=f, nthreads=4)
    random_case(s)[0].run_oracle(nthreads=4)
    random_case(s, grids=True)[0].run_oracle(nthreads=4)
print(f"oracle under ASan+UBSan(+float-cast-overflow): {fails} failed of 240 checks, 180 whole frames rendered")
sys.exit(1 if fails else 0)
PY
EXA_HIP_LIB="$D/libexa_hip.so" LD_PRELOAD="$CLANG_RT" python3 - <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, '..')
import fuzz_prep
fails = sum(bool(fuzz_prep.check(s)[0]) for s in range(600))
print(f"host preparation (exa_prep.cpp) under ASan+UBSan: {fails} failed of 600")
# ... and with cells missing (ALLOW_EMPTY_CELLS): prep == oracle byte for byte on 100 scenes with holes
from fuzz_cases import random_case
from owlexabrick_amd import binding, scenes
from oracle import pyoracle as po
for s in range(100):
    sc = scenes.with_empty_cells(random_case(s, grids=(s % 2 == 0))[0].scene, fraction=0.2, seed=s)
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields, allow_empty_cells=True)
    P = binding.Prep(sc, num_threads=3, allow_empty_cells=True)
    fails += int(P.scalars().tobytes() != S.scalars().tobytes() or P.regions().tobytes() != S.regions().tobytes())
    P.close()
print(f"host preparation with empty cells under ASan+UBSan: {fails} failed of 100")
# ... and the leaves and links of the rope walk (csrc/exa_ropes.h through exa_prep_ropes) on 200 random scenes: boxes = domains
import numpy as np
for s in range(200):
    P = binding.Prep(random_case(s, grids=(s % 2 == 0))[0].scene, num_threads=3)
    r = P.ropes()
    reg = P.regions()
    fails += int(r["flags"] != 3 or not np.array_equal(r["boxes"][:len(reg), :3], reg["dom_lo"]) or not np.array_equal(r["boxes"][:len(reg), 3:], reg["dom_hi"]))
    P.close()
print(f"rope construction (exa_ropes.h) under ASan+UBSan: {fails} failed of 200 (cumulative)")
sys.exit(1 if fails else 0)
PY
/opt/rocm/bin/hipcc $SAN -std=c++17 -shared-libsan -o "$D/exaRender" "$ROOT/owlexabrick_amd/host/exaRender.cpp" "$ROOT/owlexabrick_amd/host/exa_host.cpp" \
    -L"$ROOT/owlexabrick_amd" -lexa_hip -Wl,-rpath,"$ROOT/owlexabrick_amd" -lpthread 2>/dev/null
LD_PRELOAD="$CLANG_RT" python3 - "$D" <<'PY'
import subprocess, sys, tempfile
import numpy as np
D = sys.argv[1]
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from fuzz_cases import random_case, _octahedron
from owlexabrick_amd import scenes
fails = 0
for seed in range(150):
    rng = np.random.default_rng(0x10AD000 + seed)
    sc = random_case(seed, grids=(seed % 3 == 0), many=(seed % 5 == 0))[0].scene
    with tempfile.TemporaryDirectory() as d:
        lo, hi = sc.bounds()
        meshes = [_octahedron(0.5 * (np.asarray(lo) + np.asarray(hi)), 2.0)] if rng.uniform() < 0.4 else None
        cfg = scenes.write_exa(sc, d, "s", meshes=meshes)
        with open(cfg, "a") as f:
            if rng.uniform() < 0.5: f.write('scalar e expr "%0 2 * 1 +"\n')
            if rng.uniform() < 0.3: f.write('value_range -1 3\n')
            if rng.uniform() < 0.3: f.write('vector m s_0.scalars s_0.scalars s_0.scalars # c\n')
        r = subprocess.run([D + "/exaRender", cfg, "--info"], capture_output=True, text=True, timeout=120)
        fails += int(r.returncode != 0 or "runtime error" in r.stderr or "ERROR: " in r.stderr)
print(f"file loaders (exa_host.cpp) under ASan+UBSan: {fails} failed of 150")
sys.exit(1 if fails else 0)
PY
