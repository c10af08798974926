"""View-sharded step on 2 and 4 CPU ranks (gloo).  The driver logic is the product's; the render / loss
functions are injected with the CPU oracle because the HIP operator has no CPU path."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_render(cam, model, pipe, bg):
    from oracle import RasterSettings, rasterize_ref
    xyz = model.get_xyz
    sp = torch.zeros_like(xyz, requires_grad=True)
    st = RasterSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                        cam.world_view_transform, cam.full_proj_transform, model.active_sh_degree, cam.camera_center)
    col, radii = rasterize_ref(xyz, sp, model.get_opacity, st, shs=model.get_features, scales=model.get_scaling,
                               rotations=model.get_rotation)
    return {"render": col, "viewspace_points": sp, "visibility_filter": radii > 0, "radii": radii}


def _scene(n_views):
    from mvs_gaussian_splatting_amd.synthetic import SceneConfig, SyntheticGaussianModel, orbit_camera
    model = SyntheticGaussianModel(300, 1, seed=0, log_scale_mean=math.log(0.1), requires_grad=True)
    cams = [orbit_camera(v, n_views, 64, 48, 50.0, 50.0) for v in range(n_views)]
    targets = [torch.rand(3, 48, 64, generator=torch.Generator().manual_seed(10 + v)) for v in range(n_views)]
    return model, cams, targets


def _worker(rank, world, port, n_views, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from mvs_gaussian_splatting_amd import dist as gdist
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from oracle import l1_loss_ref
    gdist.init_process_group(backend="gloo")
    model, cams, targets = _scene(n_views)
    out = gdist.sharded_train_step(model, cams, targets, torch.zeros(3), PipelineParams(), render_fn=_oracle_render,
                                   loss_fn=l1_loss_ref, stats_fn=None)
    q.put((rank, out["loss"], out["views"], out["local_views"], float(model._xyz.grad.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_views_partition():
    from mvs_gaussian_splatting_amd.dist import views_of_rank
    for world in (1, 2, 4, 8):
        got = sorted(v for r in range(world) for v in views_of_rank(8, r, world))
        assert got == list(range(8))
    assert views_of_rank(8, 1, 4) == [1, 5]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_views", [(2, 4), (4, 8)])
def test_multi_rank_step_matches_single_process(world, n_views):
    """world 2 with 4 views (partition [r, r+2]) and world 4 with 8 views (partition [r, r+4], the C5 shape)."""
    from mvs_gaussian_splatting_amd import dist as gdist
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    from oracle import l1_loss_ref
    model, cams, targets = _scene(n_views)
    single = gdist.sharded_train_step(model, cams, targets, torch.zeros(3), PipelineParams(), render_fn=_oracle_render,
                                      loss_fn=l1_loss_ref, stats_fn=None)
    assert single["views"] == n_views and single["world"] == 1

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=480) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    for rank, loss, views, mine, g in res:
        assert views == n_views                                        # the all-reduced view count proves the group size
        assert mine == list(range(rank, n_views, world))
        assert math.isclose(loss, res[0][1], rel_tol=1e-6)             # the all-reduced loss is identical on all ranks
        assert math.isclose(loss, single["loss"], rel_tol=1e-5)        # and equals the single-process mean
        assert g > 0
