"""strong-scaling rehearsal on one GPU: kernel time of shard (rank, world) of the bench frame for
world = 1, 2, 4, 8 (the scene is replicated, so a rank's launch is exactly what it would run on its own GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
import numpy as np
sc = scenes.config("c4_exajet", scale=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
opts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4]
for order in opts:
    R.setOption("tile_order", order)
    base = None
    for world in (1, 2, 4, 8):
        ms = []
        for rank in range(world):
            R.setShard(rank, world)
            R.render(); R.render()
            t = [0, 0, 0]
            for k in range(3):
                R.render(); t[k] = R.stats()["kernel_ms"]
            ms.append(float(np.median(t)))
        if world == 1: base = ms[0]
        print(f"tile_order {order} world {world}: max {max(ms):.3f} ms  mean {np.mean(ms):.3f}  ideal {base / world:.3f}  "
              f"efficiency {base / world / max(ms):.3f}", flush=True)
