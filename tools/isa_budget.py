#!/usr/bin/env python3
"""Instruction budget of one kernel from source-annotated device assembly.

  hipcc --offload-arch=gfx950 <flags of csrc/Makefile> --cuda-device-only -S -gline-tables-only exa_kernels.hip -o annot.s
  python tools/isa_budget.py annot.s <mangled kernel name> [source.hip]

Every instruction is attributed to the source line of the nearest preceding `.loc` (the line of the innermost inlined
frame), source lines are mapped to the phases of the kd march (brick visit / sample epilogue / kd walk step / segment
pop / ray set-up + output) by the function they lie in, and instructions are classed by opcode.  Static counts: one
trip through each phase's straight-line code, not weighted by how often a lane takes it."""
import collections
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
# the source the assembly was compiled from (line numbers must match); default: the tree's exa_kernels.hip
import os
src_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "owlexabrick_amd",
                                                               "csrc", "exa_kernels.hip")
src = open(src_path).read().split("\n")

# source line -> enclosing top-level device function (crude: last line at column 0 that opens a function)
func_of_line = {}
cur = "?"
for i, l in enumerate(src, 1):
    m = re.match(r"^(?:template.*\n)?(?:__device__|__global__|static|hipError_t|template)", l)
    if re.match(r"^__device__|^__global__", l) or (l.startswith("template") and False):
        mm = re.search(r"(\w+)\s*\(", l)
        if mm:
            cur = mm.group(1)
    if l.startswith("__global__") or l.startswith("__device__"):
        pass
    func_of_line[i] = cur
# functions declared on the line after `template <...>`
cur = "?"
for i, l in enumerate(src, 1):
    if re.match(r"^(__device__|__global__)", l):
        mm = re.search(r"\b(\w+)\s*\(", l.split("__forceinline__")[-1] if "__forceinline__" in l else l)
        if mm:
            cur = mm.group(1)
    func_of_line[i] = cur


def phase_of(line):
    f = func_of_line.get(line, "?")
    if f in ("addBasisFast", "addBasisFunctions", "loadPair", "rayAt"):
        return "brick visit"
    if f in ("lookupXF", "shadeSample", "compositeSample", "integrateVolume", "fdiv", "fsqrt", "dotF", "gradOf"):
        return "sample epilogue"
    if f in ("kdStep", "kdPop", "ropeStep", "refinedRcp", "divByRcp", "ropeRayInRange"):
        return "kd walk"
    if f in ("firstSampleT", "firstSampleTPow2"):
        return "segment pop"
    if f.startswith("renderFrameKdKernel") or f == "__launch_bounds__":
        return None            # decided by line ranges inside the kernel body
    if f in ("boxTest", "xfmPoint", "xfmVector", "normalize", "length", "dot", "mk", "init", "next", "linear_to_srgb",
             "make_8bit", "make_rgba8", "clockHeat"):
        return "set-up/output"
    return "other:" + f


# line ranges inside renderFrameKdKernel, found by their marker comments
def find(marker, start=0):
    for i in range(start, len(src)):
        if marker in src[i]:
            return i + 1
    raise SystemExit("marker not found: " + marker)


k0 = find("void renderFrameKdKernel(const RenderArgs a)")
L_refill = find("// ---- refill burst", k0)
L_pop = find("// ---- next segment from this lane's queue", k0)
L_brick = find("// ---- one brick visit ----", k0)
L_final = find("// ---- all bricks of the region seen", k0)
L_endstep = find("// ---- end of this step", k0)
L_after = find("C.lap(ST_T_OTHER);", k0)
L_end = find("// Wide march: L lanes per ray", k0)


def kernel_phase(line):
    if line < L_refill:
        return "set-up/output"
    if line < L_pop:
        return "kd walk"
    if line < L_brick:
        return "segment pop"
    if line < L_final:
        return "brick visit"
    if line < L_after:
        return "sample epilogue"
    return "set-up/output"


def klass(op):
    if op.startswith("v_cvt") or op in ("v_floor_f32", "v_ceil_f32", "v_trunc_f32", "v_rndne_f32", "v_fract_f32"):
        return "cvt/floor"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "transcendental"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "compare"
    if op.startswith(("v_cndmask",)):
        return "select"
    if op.startswith(("v_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane", "v_swap")):
        return "move"
    if re.match(r"v_(add|sub|subrev|mul|fma|mac|fmac|mad|max|min|med3|max3|min3|ldexp|frexp|div_|rndne)\w*_(f32|f64|legacy_f32)", op) or op.startswith("v_pk_"):
        return "f32 arithmetic"
    if op.startswith("v_"):
        return "int/address/logic"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "wait/nop"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "scalar memory"
    if op.startswith("s_"):
        return "SALU"
    return "other"


lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
cur_line, counts, total = None, collections.defaultdict(collections.Counter), collections.Counter()
file_of_kernels = None
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        # only lines of exa_kernels.hip (file 0) move the attribution: code inlined from the HIP headers (floorf, fminf,
        # __shfl ...) stays with the line that called it
        # line 0 = code the compiler merged or moved; the small vector helpers at the top of the file (V3 operators,
        # dot, normalize, fdiv, fsqrt: lines < 62 except the LCG) are inlined everywhere: both stay with the phase
        # of the instruction before them
        ln_ = int(m.group(2))
        helper = ln_ < 70 and not (40 <= ln_ <= 62)
        if int(m.group(1)) == 0 and ln_ > 0 and not helper:
            cur_line = (0, ln_)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if not re.match(r"^[a-z]", op):
        continue
    ln = cur_line[1] if cur_line else 0
    ph = phase_of(ln)
    if ph is None:
        ph = kernel_phase(ln)
    counts[ph][klass(op)] += 1
    total[klass(op)] += 1

classes = ["f32 arithmetic", "int/address/logic", "compare", "select", "cvt/floor", "transcendental", "move",
           "vector memory", "LDS", "SALU", "scalar memory", "branch", "wait/nop", "other"]
phases = ["brick visit", "sample epilogue", "kd walk", "segment pop", "set-up/output"] + sorted(p for p in counts if p.startswith("other"))
w = max(len(p) for p in phases) + 2
print(f"kernel {kernel}")
print(" " * w + "".join(f"{c[:11]:>12s}" for c in classes) + f"{'VALU':>8s}{'all':>8s}")
valu_classes = classes[:7]
for p in phases:
    c = counts.get(p, {})
    print(f"{p:<{w}s}" + "".join(f"{c.get(k, 0):12d}" for k in classes) + f"{sum(c.get(k, 0) for k in valu_classes):8d}{sum(c.values()):8d}")
print(f"{'total':<{w}s}" + "".join(f"{total.get(k, 0):12d}" for k in classes) + f"{sum(total.get(k, 0) for k in valu_classes):8d}{sum(total.values()):8d}")
