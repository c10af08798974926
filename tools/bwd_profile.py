"""Where the cycles of render_bwd go: needs a library built with -DBWD_PROFILE (tools/build_variant.sh bwd_prof
mvs_gaussian_splatting_amd/csrc/render.hip -DBWD_PROFILE) loaded through GSR_LIB_PATH.  The kernel then sums shader-clock
cycles per section over all waves: staging of a round, list build of a pass, walk, row sums.

    GSR_LIB_PATH=$PWD/tools/ab/bwd_prof.so PYTHONPATH=.:tools python tools/bwd_profile.py [C4|C3|C2|heavy] [iters]
"""
import ctypes as C
import math
import sys

import torch

from mvs_gaussian_splatting_amd.synthetic import make_heavy_tail_model
from scene_gpu import GpuScene

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def mutate(model):
    ht = make_heavy_tail_model(model._xyz.shape[0], model.max_sh_degree, seed=3, log_footprint_mean=math.log(0.0013))
    model._xyz, model._scaling, model._opacity = ht._xyz, ht._scaling, ht._opacity


s = GpuScene("C4", P=6_000_000, mutate=mutate, fused=True) if cfg == "heavy" else GpuScene(cfg, fused=True)
dL = torch.sign(torch.rand(3, s.H, s.W, device=s.dev) - 0.5) / (3 * s.H * s.W)
for _ in range(3):
    s.forward(); s.backward(dL)
torch.cuda.synchronize()
fn = s.lib.gsr_debug_bwd_profile
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 8)()
assert fn(out, 1) == 0
for _ in range(iters):
    s.forward(); s.backward(dL)
assert fn(out, 1) == 0
v = [x / iters for x in out]
tot = v[7]
names = ["round staging (mask, record, prefetch)", "pass: counters + list build", "walk", "row sums + row write"]
print(f"{cfg}: rounds {v[4]:.0f}  passes {v[5]:.0f} ({v[5] / max(v[4], 1):.3f} per round)  steps {v[6]:.0f} ({v[6] / max(v[4], 1):.2f} per round)")
for k in range(4):
    print(f"  {names[k]:42s} {v[k] / tot * 100:5.1f} %   {v[k] / max(v[4], 1):8.0f} cycles per round")
print(f"  {'other (tile prologue / epilogue)':42s} {(tot - sum(v[:4])) / tot * 100:5.1f} %")
print(f"  walk: {v[2] / max(v[6], 1):.0f} cycles per step; wave-cycles per frame {tot:.3e}")
