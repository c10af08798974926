"""CPU oracle for the differentiable Gaussian rasterizer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product package (``mvs_gaussian_splatting_amd``) never imports this module and
fails loudly when its HIP library is missing.

PARITY STATUS: **unpinned at the rasterizer boundary**.  The reference's CUDA
rasterizer (``diff_gaussian_rasterization``, ``.gitmodules:4-6``) is an
un-vendored submodule that is absent from ``/root/reference`` and the reference
holds no tests or golden images.  The restatement therefore follows the
behavioural spec in ``SURVEY.md`` Appendix A and is pinned only on the sub-steps
for which the reference ships importable Python (``utils/sh_utils.py``,
``utils/graphics_utils.py``, ``utils/loss_utils.py``): see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``.

Two independent restatements live here and are cross-checked against each other
(``tests/test_oracle_selfcheck.py::test_c_oracle_agrees_with_pytorch_oracle``):

* ``rasterizer_ref.py`` -- vectorised PyTorch, dtype-generic, differentiable (gradient truth in float64, and the
  "pure-PyTorch CPU rasterizer" baseline of ``bench.py``);
* ``c/gsr_oracle.c`` (+ ``c_oracle.py``) -- plain C, scalar per-pixel loops in upstream's order, forward only.
"""
from .rasterizer_ref import (  # noqa: F401
    RasterSettings,
    preprocess_ref,
    bin_ref,
    render_tiles_ref,
    rasterize_ref,
    eval_sh_ref,
    build_cov3d_ref,
    l1_loss_ref,
)
