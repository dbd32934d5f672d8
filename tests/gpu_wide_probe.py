"""critical-path probe for the wide march: the tiles of rank 0 of 64 of the bench frame (a shard that is far from filling
the GPU, so its time is the longest rays' latency), marched with 1 / 2 / 4 / 8 lanes per ray"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
import numpy as np
sc = scenes.config("c4_exajet", scale=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 64      # 64: 256 tiles (one round of workgroups at 4 lanes per ray); 256: 64 tiles (one round at 16)
R.setShard(0, shards)
ref = None
for lanes in (0, 2, 4):
    R.setOption("wide_march", lanes)
    R.setOption("tile_feedback", 0)
    img = R.render()
    R.render()
    t = []
    for _ in range(5):
        R.render(); t.append(R.stats()["kernel_ms"])
    same = True if ref is None else bool(np.array_equal(img, ref))
    if ref is None:
        ref = img.copy()
    print(f"critical-path probe (rank 0 of {shards}), lanes per ray {max(lanes, 1)}: {np.median(t):.3f} ms  (pixels identical: {same})", flush=True)
# work ratio: the whole 1024x1024 frame with every tile wide (throughput-bound), EXA_WIDE_BUDGET_GB must admit all tiles
R.setShard(0, 1)
R.resizeFrameBuffer((1024, 1024))
cam = __import__("owlexabrick_amd.harness", fromlist=["x"]).default_camera(*R.voxelSpaceBounds, 1024, 1024)
R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
base = None
for lanes in (0, 2, 4):
    R.setOption("wide_march", lanes)
    R.render(); R.render()
    t = []
    for _ in range(5):
        R.render(); t.append(R.stats()["kernel_ms"])
    if base is None:
        base = np.median(t)
    print(f"whole 1024^2 frame, lanes per ray {max(lanes, 1)}: {np.median(t):.3f} ms  (work x{np.median(t) / base:.2f})", flush=True)
