"""Fused loss kernels at 1080p: L1 and L1 + D-SSIM (value + gradient), next to the torch graph of utils/loss_utils.py."""
import time
import torch
import torch.nn.functional as F
from mvs_gaussian_splatting_amd import l1_loss, l1_dssim_loss

dev = torch.device("cuda:0")
x = torch.rand(3, 1080, 1920, device=dev, requires_grad=True)
y = torch.rand(3, 1080, 1920, device=dev)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fused(loss_fn):
    def f():
        x.grad = None
        loss_fn(x, y).backward()
    return f


def torch_ref():
    g = torch.tensor([torch.exp(torch.tensor(-(i - 5) ** 2 / (2 * 1.5 ** 2))) for i in range(11)], device=dev)
    g = (g / g.sum()).unsqueeze(1)
    w = (g @ g.t()).expand(3, 1, 11, 11).contiguous()

    def f():
        x.grad = None
        a, b = x.unsqueeze(0), y.unsqueeze(0)
        mu1, mu2 = F.conv2d(a, w, padding=5, groups=3), F.conv2d(b, w, padding=5, groups=3)
        s1 = F.conv2d(a * a, w, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(b * b, w, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a * b, w, padding=5, groups=3) - mu1 * mu2
        ssim = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        loss = 0.8 * (x - y).abs().mean() + 0.2 * (1 - ssim.mean())
        loss.backward()
    return f


print(f"fused L1            : {timeit(fused(l1_loss)):.3f} ms (value + gradient)")
print(f"fused L1 + D-SSIM   : {timeit(fused(lambda a, b: l1_dssim_loss(a, b, 0.2))):.3f} ms (value + gradient)")
print(f"torch graph (reference's formulation, MIOpen convs): {timeit(torch_ref(), 10):.3f} ms")
