"""Extra fuzz seeds of tests/test_gpu_parity.py::test_fuzz_random_small_scenes_forward_and_backward (not part of the
suite: a one-off soak after kernel changes):  python tools/fuzz_more.py 8 48"""
import os
import sys
import traceback

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_parity as t  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
bad, skipped = [], []
for seed in range(lo, hi):
    try:
        t.test_fuzz_random_small_scenes_forward_and_backward(dev, seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as ex:
        if "too ill-conditioned to test" in str(ex):      # tests/grad_util.py ESCAPE_CAP: the float32 oracle itself is > 1e-4
            skipped.append(seed)                           # off float64 on this scene; nothing can be concluded from it
            print("seed", seed, "SKIPPED (ill-conditioned scene):", str(ex)[:300], flush=True)
        elif "the mask must leave most of the image in the loss" in str(ex):
            # decided by the float64 ORACLE alone, before the HIP gradients are looked at: more than 30 % of the loss
            # weights are zero (seed 418: 1069 splats of sigma 2.4 px on 142 x 126 pixels -- sixty tails per pixel, two
            # thirds of the pixels have some alpha within 1e-4 of 1/255).  Lists, radii and pixels of the scene WERE
            # checked above; the masked gradient comparison has nothing left to compare.  Counted with the refused scenes.
            skipped.append(seed)
            print("seed", seed, "SKIPPED (oracle masks more than 30 % of the loss):", str(ex)[:200], flush=True)
        else:
            bad.append(seed)
            print("seed", seed, "FAILED", flush=True)
            traceback.print_exc()
    except Exception:      # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED", flush=True)
        traceback.print_exc()
print("skipped (ill-conditioned / mostly masked):", skipped)
print("failed seeds:", bad)
n = max(hi - lo, 1)
if len(skipped) > max(2, n // 20):
    print(f"too many scenes refused as ill-conditioned ({len(skipped)} of {n}): the soak proves less than it claims")
    sys.exit(2)
sys.exit(1 if bad else 0)
