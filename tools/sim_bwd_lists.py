"""Offline estimate (CPU, oracle preprocess) of the step counts of list-driven backward walks: rounds of 64 instances
from the back of a tile's list, the round costs max(list lengths) steps.  Compares four 8x8 sub-block lists (the round-3
kernel) with sixteen 4x4 mini-block lists.  Usage: python tools/sim_bwd_lists.py [C4|C3|C2] [tiles] [instances per round]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from mvs_gaussian_splatting_amd.synthetic import CONFIGS, make_scene
import math
from oracle import rasterizer_ref as R

ROUND = int(sys.argv[3]) if len(sys.argv) > 3 else 64      # instances per round


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    ntiles = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    cfg = CONFIGS[name]
    model, cam, bg, _ = make_scene(cfg)
    st = R.RasterSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                          cam.world_view_transform, cam.full_proj_transform, cfg.sh_degree, cam.camera_center)
    with torch.no_grad():
        pre = R.preprocess_ref(model.get_xyz, model.get_opacity, st, shs=model.get_features, scales=model.get_scaling,
                               rotations=model.get_rotation)
    keys, vals, ranges = R.bin_ref(pre)
    idx = pre["idx"].numpy(); inv = np.full(model.get_xyz.shape[0], -1, np.int64); inv[idx] = np.arange(idx.size)
    xy = pre["v_xy"].numpy(); con = pre["v_conic"].numpy(); op = pre["v_opacity"].numpy()
    gx, gy = pre["grid"]
    rng = np.random.default_rng(0)
    tiles = rng.choice(gx * gy, ntiles, replace=False)
    tot = dict(inst=0, rounds=0, s4=0, s16=0, p4=0, p16=0, alive=0, maxpop=0)
    yy, xx = np.mgrid[0:16, 0:16]
    for t in tiles:
        a, b = ranges[t]
        if b <= a: continue
        v = inv[vals[a:b]]
        tx, ty = (t % gx) * 16, (t // gx) * 16
        dx = xy[v, 0][:, None, None] - (tx + xx)[None]; dy = xy[v, 1][:, None, None] - (ty + yy)[None]
        power = -0.5 * (con[v, 0][:, None, None] * dx * dx + con[v, 2][:, None, None] * dy * dy) - con[v, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, op[v][:, None, None] * np.exp(power))
        ok = (alpha >= 1 / 255) & (power <= 0)
        m4 = ok.reshape(-1, 2, 8, 2, 8).any(axis=(2, 4)).reshape(-1, 4)
        m16 = ok.reshape(-1, 4, 4, 4, 4).any(axis=(2, 4)).reshape(-1, 16)
        keep = m4.any(1)
        m4, m16, ok = m4[keep], m16[keep], ok[keep]
        n = m4.shape[0]
        tot["inst"] += n
        for hi in range(n, 0, -ROUND):
            lo = max(0, hi - ROUND)
            tot["rounds"] += 1
            tot["s4"] += m4[lo:hi].sum(0).max(); tot["p4"] += m4[lo:hi].sum()
            tot["s16"] += m16[lo:hi].sum(0).max(); tot["p16"] += m16[lo:hi].sum()
            tot["maxpop"] += m16[lo:hi].sum(1).max()
        tot["alive"] += ok.sum()
    r = tot["rounds"]
    print(f"{name}: {tot['inst']} instances in {ntiles} tiles, {r} rounds")
    print(f"  4 lists (8x8):  pairs/round {tot['p4']/r:.1f}  mean list {tot['p4']/r/4:.1f}  steps/round {tot['s4']/r:.2f}  padding {4*tot['s4']/tot['p4']-1:.3f}  lane eff {tot['alive']/(64*tot['p4']):.3f}")
    print(f" 16 lists (4x4):  pairs/round {tot['p16']/r:.1f}  mean list {tot['p16']/r/16:.1f}  steps/round {tot['s16']/r:.2f}  padding {16*tot['s16']/tot['p16']-1:.3f}  lane eff {tot['alive']/(16*tot['p16']):.3f}")
    print(f"  steps ratio 16/4: {tot['s16']/tot['s4']:.3f}   mean over rounds of max popcount(m16): {tot['maxpop']/r:.1f}")
main()
