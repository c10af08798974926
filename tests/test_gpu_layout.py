"""The Morton layout step on the GPU: a frame of the reordered model is the frame of the original one -- image, radii,
every gradient and the densification statistics, bit for bit, rows permuted -- when no two Gaussians of one tile share
a float32 depth (ties are broken by index, as in the reference's stable sort, so a tie may blend in the other order).
The full-size effect on the time is ``tools/bench_layout.py`` (``profiles/r04/layout_morton.txt``)."""
import pytest
import torch

from conftest import small_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [2, 7])
def test_reordered_model_renders_and_trains_to_the_same_bits(gpu_device, seed):
    from mvs_gaussian_splatting_amd import render, l1_loss, add_densification_stats
    from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
    from mvs_gaussian_splatting_amd.synthetic import PipelineParams
    dev = gpu_device
    model, cam, bg, target = small_scene(P=3000, sh_degree=3, width=352, height=208, scale=0.03, seed=seed)
    model.to(dev); cam.to(dev)
    bg, target = bg.to(dev), target.to(dev)

    def step():
        for p in model.parameters():
            p.requires_grad_(True)
            p.grad = None
        for t in (model.xyz_gradient_accum, model.denom, model.max_radii2D):
            t.zero_()
        pkg = render(cam, model, PipelineParams(), bg)
        l1_loss(pkg["render"], target).backward()
        add_densification_stats(model, pkg["viewspace_points"], pkg["radii"])
        grads = [p.grad.detach().clone() for p in model.parameters()] + [pkg["viewspace_points"].grad.detach().clone()]
        stats = [t.clone() for t in (model.xyz_gradient_accum, model.denom, model.max_radii2D)]
        return pkg["render"].detach().clone(), pkg["radii"].clone(), pkg["visibility_filter"].clone(), grads, stats

    img0, radii0, vis0, grads0, stats0 = step()
    # the precondition: float32 view depths of the visible Gaussians are pairwise distinct in this scene
    z = (model._xyz.detach() @ cam.world_view_transform[:3, 2] + cam.world_view_transform[3, 2])[radii0 > 0]
    if z.unique().numel() != z.numel():
        pytest.skip("this seed has equal depths among its visible Gaussians")
    perm = reorder_gaussians_(model)
    assert not torch.equal(perm, torch.arange(perm.numel(), device=dev))
    img1, radii1, vis1, grads1, stats1 = step()
    assert int((radii0 > 0).sum()) > 500
    assert torch.equal(img1, img0)
    assert torch.equal(radii1, radii0[perm]) and torch.equal(vis1, vis0[perm])
    for a, b in zip(grads1, grads0):
        assert torch.equal(a, b[perm])
    for a, b in zip(stats1, stats0):
        assert torch.equal(a, b[perm])


def test_full_size_reordering_changes_nothing_but_the_order_of_equal_depth_ties(gpu_device):
    """BASELINE config 4 (6 M Gaussians, 1080p) in index order and in Morton order: the (tile, depth) keys of the sorted
    instance list are the same array; the lists name the same Gaussians except INSIDE runs of equal keys (equal float32
    depth in one tile: broken by index, as upstream's stable sort does, so the other storage order flips them); and
    every pixel that differs lies in a tile that holds such a run."""
    import numpy as np
    from gpu_util import forward_with_state, product_settings
    from mvs_gaussian_splatting_amd.layout import reorder_gaussians_
    from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
    cfg = CONFIGS["C4"]
    model, cam, bg, _ = make_scene(cfg)
    model.to(gpu_device); cam.to(gpu_device)
    st = product_settings(cam, bg, cfg.sh_degree, gpu_device)

    def lists():
        with torch.no_grad():
            o = forward_with_state(gpu_device, st, model.get_xyz, model.get_opacity, shs=model.get_features,
                                   scales=model.get_scaling, rotations=model.get_rotation, binning_mode=0)
        return o["keys"], o["point_list"].astype(np.int64), o["color"], o["radii"]

    keys0, plist0, img0, radii0 = lists()
    perm = reorder_gaussians_(model).cpu().numpy()
    keys1, plist1, img1, radii1 = lists()
    assert np.array_equal(keys0, keys1)
    assert torch.equal(radii1, radii0[torch.from_numpy(perm)])
    back = perm[plist1]                                       # the Morton-order list in the original numbering
    differ = back != plist0
    tie = np.zeros(keys0.size, dtype=bool)
    eq = keys0[1:] == keys0[:-1]
    tie[1:] |= eq
    tie[:-1] |= eq
    assert not np.any(differ & ~tie)                          # outside equal-key runs: the same Gaussian at the same place
    # inside a run the same SET of Gaussians: sort both lists within equal keys and compare
    order0 = np.lexsort((plist0, keys0))
    order1 = np.lexsort((back, keys1))
    assert np.array_equal(plist0[order0], back[order1])
    gx = (cfg.width + 15) // 16
    tie_tiles = np.unique((keys0[tie] >> np.uint64(32)).astype(np.int64))
    diff_px = (img0 != img1).any(dim=0).numpy()
    ys, xs = np.nonzero(diff_px)
    px_tiles = np.unique((ys // 16) * gx + xs // 16)
    assert np.all(np.isin(px_tiles, tie_tiles))
    print(f"[layout C4] equal-key entries {int(tie.sum())} in {tie_tiles.size} tiles; list entries in another order "
          f"{int(differ.sum())}; pixels that differ {int(diff_px.sum())} in {px_tiles.size} tiles; "
          f"max |difference| {float((img0 - img1).abs().max()):.2e}")
