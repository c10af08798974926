"""Work counters of the forward compositing kernel on a bench config (tuning aid, GPU only)."""
import ctypes as C
import sys

import torch

from mvs_gaussian_splatting_amd import _lib
from scene_gpu import GpuScene

s = GpuScene(sys.argv[1] if len(sys.argv) > 1 else "C4")
s.forward()
stats = torch.zeros(8, dtype=torch.int64, device=s.dev)
_lib.check(s.lib.gsr_debug_render_stats(C.byref(s.params), s.geom.data_ptr(), s.binning.data_ptr(), s.img.data_ptr(),
                                        s.R, s.V, s.color.data_ptr(), stats.data_ptr(), s.stream), "stats")
torch.cuda.synchronize()
v = stats.cpu().tolist()
print(dict(config=s.cfg.name, R=s.R, visible=int((s.radii > 0).sum()), in_lists=v[0], staged=v[1], visited=v[2],
           evals=v[3], evals_with_hit=v[4], sum_tile_max=v[5]))
