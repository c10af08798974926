"""Shared setup for the GPU tuning tools: a bench scene resident on cuda:0 plus raw C-ABI handles."""
import ctypes as C
import math

import torch

from mvs_gaussian_splatting_amd import _lib
from mvs_gaussian_splatting_amd.rasterizer import GaussianRasterizationSettings, _make_params, _counts_pinned_thread
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene


class GpuScene:
    def __init__(self, cfgname="C4", P=None, view=0, mutate=None, fused=False):
        self.cfg = cfg = CONFIGS[cfgname]
        self.dev = dev = torch.device("cuda:0")
        self.lib = _lib.load()
        model, cam, bg, target = make_scene(cfg, P=P, view=view)
        if mutate is not None:
            mutate(model)          # reshape the cloud (CPU tensors) before it moves to the device
        model.to(dev); cam.to(dev)
        self.model, self.cam, self.bg, self.target = model, cam, bg.to(dev), target.to(dev)
        self.W, self.H, self.P = cfg.width, cfg.height, P or cfg.P
        self.settings = GaussianRasterizationSettings(
            self.H, self.W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), self.bg, 1.0,
            cam.world_view_transform, cam.full_proj_transform, cfg.sh_degree, cam.camera_center, False, False)
        e = torch.empty(0, device=dev)
        with torch.no_grad():
            # keep every input alive for as long as `params` (raw pointers) is used
            self.inputs = [model.get_xyz.contiguous(), model.get_features.contiguous(), e,
                           model.get_opacity.contiguous(), model.get_scaling.contiguous(),
                           model.get_rotation.contiguous(), e]
        self.fused = fused
        if fused:   # raw parameters, split SH, activations inside the kernels (what render() feeds by default)
            self.inputs = [model._xyz.contiguous(), model._features_dc.contiguous(), e, model._opacity.contiguous(),
                           model._scaling.contiguous(), model._rotation.contiguous(), e, model._features_rest.contiguous()]
            self.params, self.keep = _make_params(dev, self.settings, *self.inputs[:7], sh_rest=self.inputs[7],
                                                  act_flags=_lib.ACT_SCALE_EXP | _lib.ACT_ROT_NORMALIZE | _lib.ACT_OPACITY_SIGMOID)
        else:
            self.params, self.keep = _make_params(dev, self.settings, *self.inputs)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.pinned = _counts_pinned_thread()[0]
        self.params.counts_pinned = self.pinned.data_ptr()
        lib = self.lib
        self.geom = torch.empty(lib.gsr_geom_bytes(self.P), dtype=torch.uint8, device=dev)
        self.img = torch.empty(lib.gsr_image_bytes(self.W, self.H), dtype=torch.uint8, device=dev)
        self.radii = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.color = torch.empty(3, self.H, self.W, device=dev)
        self.R = 0
        self.binning = None

    def forward(self):
        lib = self.lib
        R, V = C.c_uint32(0), C.c_uint32(0)
        _lib.check(lib.gsr_forward_preprocess(C.byref(self.params), self.geom.data_ptr(), self.radii.data_ptr(),
                                              self.stream, C.byref(R), C.byref(V)), "pre")
        self.R, self.V = R.value, V.value
        nb = lib.gsr_binning_bytes(self.R, self.V, self.W, self.H, self.params.binning_mode)
        if self.binning is None or self.binning.numel() < nb:
            self.binning = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        _lib.check(lib.gsr_forward_render(C.byref(self.params), self.geom.data_ptr(), self.binning.data_ptr(),
                                          self.binning.numel(), self.img.data_ptr(), self.R, self.V, self.color.data_ptr(),
                                          self.stream), "render")

    def forward_sync_free(self, capacity=None, event=None):
        """gsr_forward with a caller-side capacity (default 1.5 x the last two-call frame's count); event: a handle from
        gsr_event_create recorded behind the scan kernel, as the operator's verified mode does."""
        lib = self.lib
        cap = int(capacity or (self.cap if hasattr(self, "cap") else int(self.R * 1.5)))
        self.cap = cap
        nb = lib.gsr_binning_bytes(cap, self.P, self.W, self.H, self.params.binning_mode)
        if self.binning is None or self.binning.numel() < nb:
            self.binning = torch.empty(nb, dtype=torch.uint8, device=self.dev)
        _lib.check(lib.gsr_forward(C.byref(self.params), self.geom.data_ptr(), self.binning.data_ptr(), self.binning.numel(),
                                   cap, self.img.data_ptr(), self.radii.data_ptr(), self.color.data_ptr(), event,
                                   self.stream), "gsr_forward")
        self.R, self.V = cap, self.P      # what the workspaces are laid out for (gsr_backward takes these)

    def backward(self, dL_dpix):
        lib, dev, P = self.lib, self.dev, self.P
        if not hasattr(self, "grads"):
            new = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
            if self.fused:
                self.g = [new(P, 3), new(P, 3), new(P, 1, 3), new(P, 1), new(P, 3), new(P, 4), new(P, 15, 3)]
                rest = self.g[6].data_ptr()
            else:
                self.g = [new(P, 3), new(P, 3), new(P, 16, 3), new(P, 1), new(P, 3), new(P, 4)]
                rest = None
            st = [None, None, None]
            if getattr(self, "fuse_stats", False):     # densification statistics taken by preprocess_bwd's epilogue
                self.stats = [torch.zeros(P, device=dev) for _ in range(3)]
                st = [t.data_ptr() for t in self.stats]
            self.grads = _lib.GsrGrads(self.g[0].data_ptr(), self.g[1].data_ptr(), self.g[2].data_ptr(), None,
                                       self.g[3].data_ptr(), self.g[4].data_ptr(), self.g[5].data_ptr(), None, rest, *st)
        nb = lib.gsr_backward_bytes(P, self.R)
        if not hasattr(self, "bwd_ws") or self.bwd_ws.numel() < nb:
            self.bwd_ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.gsr_backward(C.byref(self.params), self.radii.data_ptr(), self.geom.data_ptr(),
                                    self.binning.data_ptr(), self.img.data_ptr(), self.R, self.V, dL_dpix.data_ptr(),
                                    self.bwd_ws.data_ptr(), self.bwd_ws.numel(), C.byref(self.grads), self.stream), "bwd")
