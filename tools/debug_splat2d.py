"""Per-parameter gradient error of the HIP config-1 path and of the float32 oracle against the float64 oracle."""
import sys
import torch
sys.path.insert(0, "tests")
from test_gpu_splat2d import _hip_step, _oracle_step, _random_inputs, NAMES

dev = torch.device("cuda:0")
for K, size, N in [(32, (96, 80, 3), 150), (64, (64, 64, 3), 100), (17, (128, 128, 3), 300), (101, (128, 128, 3), 257),
                   (5, (33, 47, 3), 1)]:
    ins, gen = _random_inputs(K * 7 + N, N, sig=(0.05, 1.0) if K > 8 else (1.5, 3.0))
    dL = torch.randn(size[0], size[1], 3, generator=gen)
    img, _, grads = _hip_step(dev, K, ins, size, dL=dL)
    img32, _, g32 = _oracle_step(K, ins, size, torch.float32, dL=dL)
    img64, _, g64 = _oracle_step(K, ins, size, torch.float64, dL=dL)
    print(f"K={K} size={size} N={N}: img hip-32 {float((img-img32).abs().max()):.2e} hip-64 {float((img.double()-img64).abs().max()):.2e} 32-64 {float((img32.double()-img64).abs().max()):.2e}")
    for a, b, c, k in zip(grads, g32, g64, NAMES):
        m = float(c.abs().max())
        eh = (a.double() - c).abs().reshape(len(c), -1).max(dim=1).values
        eo = (b.double() - c).abs().reshape(len(c), -1).max(dim=1).values
        i = int(eh.argmax())
        print(f"   {k:8s} max|g| {m:.3e}  hip err {float(eh.max())/m:.2e} (at {i}: sx {float(ins[0][i]):.3f} sy {float(ins[1][i]):.3f} rho {float(ins[2][i]):.3f})  fp32-oracle err {float(eo.max())/m:.2e}")
