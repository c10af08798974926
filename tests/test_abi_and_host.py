"""C-ABI surface and host logic without a GPU: the library loads, exports every symbol include/gsr.h
declares, struct layouts agree with a C compiler, and the operator refuses to run without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsr.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mvs_gaussian_splatting_amd import _lib
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gsr.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature in _lib.SYMBOLS"
    assert lib.gsr_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.gsr_build_info()


def test_struct_layouts_match_the_c_compiler():
    from mvs_gaussian_splatting_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "gsr.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %d %zu %zu %d %zu %zu\n", sizeof(GsrParams), sizeof(GsrGrads), offsetof(GsrParams, means3D),
         offsetof(GsrParams, bg), offsetof(GsrParams, counts_pinned), offsetof(GsrGrads, dL_dshs_rest), GSR_STAGE_COUNT,
         offsetof(GsrParams, debug_flags), offsetof(GsrGrads, stats_max_radii2D), GSR_ABI_VERSION,
         offsetof(GsrParams, visible_out), offsetof(GsrParams, depth_span_lt24));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    vals = list(map(int, out))
    P, G = _lib.GsrParams, _lib.GsrGrads
    assert vals == [C.sizeof(P), C.sizeof(G), P.means3D.offset, P.bg.offset, P.counts_pinned.offset, G.dL_dshs_rest.offset,
                    _lib.STAGE_COUNT, P.debug_flags.offset, G.stats_max_radii2D.offset, _lib.ABI_VERSION,
                    P.visible_out.offset, P.depth_span_lt24.offset]


def test_workspace_sizes_are_monotone_and_aligned():
    from mvs_gaussian_splatting_amd import _lib
    lib = _lib.load()
    assert lib.gsr_geom_bytes(0) % 256 == 0
    assert lib.gsr_geom_bytes(1000) > 64 * 1000
    assert lib.gsr_geom_bytes(2000) > lib.gsr_geom_bytes(1000)
    assert lib.gsr_image_bytes(1920, 1080) >= 8 * 1920 * 1080
    assert lib.gsr_binning_bytes(10_000_000, 4_000_000, 1920, 1080, 1) >= 24 * 10_000_000
    # two-level binning: two (tile, index) buffers per instance; the per-Gaussian payload lives in the geometry workspace
    assert lib.gsr_binning_bytes(10_000_000, 4_000_000, 1920, 1080, 0) >= 16 * 10_000_000
    assert lib.gsr_geom_bytes(1000) > (64 + 16 + 2 * 4 + 2 * 8) * 1000
    assert lib.gsr_backward_bytes(1000, 5000) >= 37 * 5000      # 36-byte row + flag byte per instance
    assert all(lib.gsr_stage_name(i) for i in range(_lib.STAGE_COUNT))


def test_bad_arguments_are_rejected_before_any_launch():
    from mvs_gaussian_splatting_amd import _lib
    lib = _lib.load()
    p = _lib.GsrParams()
    p.P, p.width, p.height = 10, 64, 64
    n = C.c_uint32(0)
    rc = lib.gsr_forward_preprocess(C.byref(p), None, None, None, C.byref(n), C.byref(n))
    assert rc == -1 and b"non-NULL" in lib.gsr_last_error()
    with pytest.raises(_lib.GsrError):
        _lib.check(rc, "gsr_forward_preprocess")
    assert lib.gsr_forward_preprocess(None, None, None, None, C.byref(n), C.byref(n)) == -1
    p.width = 0
    assert lib.gsr_forward_preprocess(C.byref(p), None, None, None, C.byref(n), C.byref(n)) == -1
    # the sync-free entry point: same validation, plus its own preconditions (all refused before any HIP call)
    assert lib.gsr_forward(None, None, None, 0, 1, None, None, None, None, None) == -1
    p.width = 64
    assert lib.gsr_forward(C.byref(p), None, None, 0, 1, None, None, None, None, None) == -1
    assert lib.gsr_event_wait(None) == -1 and lib.gsr_event_query(None, None) == -1
    assert lib.gsr_event_destroy(None) == 0


def _cpu_call(**over):
    from mvs_gaussian_splatting_amd import GaussianRasterizer, GaussianRasterizationSettings
    from conftest import small_scene, make_settings
    model, cam, bg, _ = small_scene(P=16, sh_degree=0, width=32, height=32)
    st = make_settings(cam, bg, 0, cls=GaussianRasterizationSettings)
    kw = dict(means3D=model.get_xyz, means2D=torch.zeros(16, 3), opacities=model.get_opacity, shs=model.get_features,
              scales=model.get_scaling, rotations=model.get_rotation)
    kw.update(over)
    return GaussianRasterizer(st)(**kw)


def test_operator_validates_inputs_like_the_reference_module():
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        _cpu_call(colors_precomp=torch.zeros(16, 3))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        _cpu_call(shs=None)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        _cpu_call(cov3D_precomp=torch.zeros(16, 6))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        _cpu_call(scales=None)


def test_no_cpu_fallback_exists():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from mvs_gaussian_splatting_amd import _lib, render, l1_loss
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from conftest import small_scene
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        _cpu_call()
    model, cam, bg, target = small_scene(P=16, sh_degree=0, width=32, height=32)
    with pytest.raises(_lib.GsrError):
        render(cam, model, PipelineParams(), bg)
    with pytest.raises(_lib.GsrError):
        l1_loss(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))
    # the fixed-model helpers too: no graph, no streams, no silent CPU rendering
    from mvs_gaussian_splatting_amd.graphed import GraphedRenderer, MultiStreamRenderer
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        GraphedRenderer(model, PipelineParams(), bg)
    with pytest.raises(_lib.GsrError, match="no CPU path"):
        MultiStreamRenderer(model, PipelineParams(), bg)
    with pytest.raises(ValueError):
        MultiStreamRenderer(model, PipelineParams(), bg, streams=0)
    # and the product package never imports the oracle
    import mvs_gaussian_splatting_amd as pkg
    pkg_dir = os.path.dirname(pkg.__file__)
    for fn in os.listdir(pkg_dir):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg_dir, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_missing_library_is_a_loud_error(monkeypatch):
    from mvs_gaussian_splatting_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgsr_hip.so")
    with pytest.raises(_lib.GsrError, match="HIP extension not built"):
        _lib.load()


def test_binning_mode_selection(monkeypatch):
    """The mode is a module setting handed to the library in GsrParams.binning_mode (GSR_BINNING only seeds it at
    import): no per-call environment reads on the hot path."""
    from mvs_gaussian_splatting_amd import _lib, rasterizer
    monkeypatch.setattr(rasterizer, "_binning_mode_value", _lib.BINNING_TWO_LEVEL_CULLED)
    for name, val in (("keys64", _lib.BINNING_KEYS64), ("two_level", _lib.BINNING_TWO_LEVEL), ("CULLED", _lib.BINNING_TWO_LEVEL_CULLED)):
        rasterizer.set_binning_mode(name)
        assert rasterizer._binning_mode_value == val
    assert rasterizer.set_binning_mode("culled") == "culled"
    with pytest.raises(ValueError):
        rasterizer.set_binning_mode("fastest")
    src = open(rasterizer.__file__).read()
    body = src[src.index("def _make_params"):]
    assert "os.environ" not in body, "no environment reads after import"
