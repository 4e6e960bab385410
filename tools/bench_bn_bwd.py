"""GPU: time of one train-mode BatchNorm(+PReLU) forward and backward at 64 images, fp32 and bf16 tensors
(srgen_bn_train_fwd / srgen_bn_train_bwd and their _bf16 namesakes: statistics, finish, apply)."""
import sys, time, torch
sys.path.insert(0, '.')
from superresolution_amd import training as T
ws = T.Workspace(torch.device("cuda"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.rand(96, device="cuda") + 0.5; b = torch.randn(96, device="cuda") * 0.1; sl = torch.rand(96, device="cuda") * 0.4
for dt in (torch.float32, torch.bfloat16):
    c = torch.randn(n, 64, 64, 96, device="cuda").to(dt); d = (torch.randn(n, 64, 64, 96, device="cuda") * 1e-3).to(dt)
    _, st = T.bn_train_fwd(c, g, b, ws, slope=sl)
    for name, fn in (("fwd", lambda: T.bn_train_fwd(c, g, b, ws, slope=sl)), ("bwd", lambda: T.bn_train_bwd(c, d, st, g, b, ws, slope=sl))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): r = fn()
        torch.cuda.synchronize(); print("bn_train_%s %s: %.1f us per call" % (name, str(dt).split('.')[-1], (time.perf_counter() - t) / 50 * 1e6))
