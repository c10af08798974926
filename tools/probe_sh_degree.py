"""Is preprocess_fwd bound by bytes or by how it asks for them?  Same C4 scene, active SH degree 3 (every visible lane reads
its 180-byte f_rest row with per-lane loads) against degree 0 (only the 12-byte DC term is read): if the kernel were
bandwidth-bound at one efficiency, the time would fall with the bytes.   PYTHONPATH=.:tools python tools/probe_sh_degree.py"""
import torch

from mvs_gaussian_splatting_amd import _lib
from scene_gpu import GpuScene

s = GpuScene("C4", fused=True)
V = None
base_mode = s.params.binning_mode
for deg, mode in ((3, base_mode), (0, base_mode), (3, 0), (0, 0), (3, base_mode), (0, base_mode)):
    s.params.D = deg
    s.params.binning_mode = mode
    s.binning = None
    for _ in range(5):
        s.forward()
    torch.cuda.synchronize()
    prof = _lib.StageProfile()
    s.params.profile = prof._h
    n = 20
    for _ in range(n):
        s.forward()
    torch.cuda.synchronize()
    res = prof.collect()
    s.params.profile = None
    V = s.V
    ms = res["preprocess_fwd"][0] / n
    sh_bytes = 180 * V if deg > 0 else 0
    total = 44 * s.P + (12 + 75) * V + sh_bytes
    print(f"binning mode {mode}, active degree {deg}: preprocess_fwd {ms:.4f} ms; model bytes {total / 1e9:.3f} GB -> {total / ms / 1e6:.0f} GB/s")
