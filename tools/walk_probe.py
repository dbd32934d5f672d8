#!/usr/bin/env python3
"""How coherent are the 64 kd STACK walks of a wave?  (GPU box; diagnostic of rounds 3-4 for the stack walk, option walk = 1)
Renders one frame of a bench configuration through the counting variant with option walk_probe = 1 and prints, per frame:
lane node steps (sum over rays), wave-level node-step executions of the per-lane walk as shipped (with their lane
utilisation), the size of the UNION of visited nodes per wave summed over the waves — the least number of node steps a walk
that takes one node at a time for the whole wave (node in scalar registers, shared stack) would need — and the restarts of
the 4-entry short stack.
    python tools/walk_probe.py [--config c4_exajet] [--scale 1.0] [--size 2048] [--camera default|closeup]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owlexabrick_amd import binding, harness, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4_exajet")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--camera", default="default")
args = ap.parse_args()
W = H = args.size
scene = scenes.config(args.config, scale=args.scale, threads=8)
prep = binding.Prep(scene, num_threads=8)
lo, hi = prep.voxel_bounds()
cam = (harness.closeup_camera if args.camera == "closeup" else harness.default_camera)(lo, hi, W, H)
xf = harness.default_xf()
R = binding.Renderer(prep)
R.resizeFrameBuffer((W, H))
R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
R.updateXF(0, xf[:, 3], xf[:, :3], scene.value_range, 1.0)
for c in range(1, len(scene.fields)):
    R.updateXF(c, xf[:, 3], xf[:, :3], (0.0, 1.0), 1.0)
R.updateIsoValues([0, 0], [0, 0], [0, 0])
R.setSpaceSkipping(True)
R.setGradientShadingDVR(True)
R.updateDt(0.5)
R.updateFrameID(0)
R.setOption("walk", 1)                  # the probe is about the stack walk (the rope walk keeps no stack and visits 1.3 records per segment)
R.setOption("walk_probe", 1)
_, st = R.renderStats()
waves = ((W + 15) // 16) * ((H + 15) // 16) * 4
d = st["diag"]
out = {"config": args.config, "scale": args.scale, "size": W, "camera": args.camera, "waves": waves,
       "segments": st["segments"], "lane_node_steps": st["nodes_visited"],
       "wave_node_step_executions_as_shipped": d[4], "lanes_per_execution": d[5] / max(1, d[4]),
       "union_nodes_summed_over_waves": st["walk_union_nodes"], "probe_overflow": st["walk_probe_overflow"],
       "union_over_shipped_executions": st["walk_union_nodes"] / max(1, d[4]),
       "lane_steps_per_union_node": st["nodes_visited"] / max(1, st["walk_union_nodes"]),
       "short_stack_restarts": st["walk_restarts"], "restarts_per_segment": st["walk_restarts"] / max(1, st["segments"]),
       "nodes_per_segment": st["nodes_visited"] / max(1, st["segments"]),
       "wave_march_iterations_summed": st["wave_iters"], "four_times_slowest_wave_per_workgroup_summed": st["tile_iters"],
       "workgroup_evenness": st["wave_iters"] / max(1, st["tile_iters"])}
print(json.dumps(out, indent=1))
