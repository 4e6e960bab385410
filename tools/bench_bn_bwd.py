"""GPU: time of one train-mode BatchNorm(+PReLU) backward at 64 images (srgen_bn_train_bwd: statistics, finish, apply)."""
import sys, time, torch
sys.path.insert(0, '.')
from superresolution_amd import training as T
ws = T.Workspace(torch.device("cuda"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c = torch.randn(n, 64, 64, 96, device="cuda"); d = torch.randn_like(c) * 1e-3
g = torch.rand(96, device="cuda") + 0.5; b = torch.randn(96, device="cuda") * 0.1; sl = torch.rand(96, device="cuda") * 0.4
_, st = T.bn_train_fwd(c, g, b, ws, slope=sl)
for _ in range(5): T.bn_train_bwd(c, d, st, g, b, ws, slope=sl)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): r = T.bn_train_bwd(c, d, st, g, b, ws, slope=sl)
torch.cuda.synchronize(); print("bn_train_bwd: %.1f us per call" % ((time.perf_counter() - t) / 50 * 1e6))
