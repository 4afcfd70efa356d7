"""f(x) at m = 8192 a few times -- run under rocprofv3 --kernel-trace --stats to split the two-level Cholesky
into its step launches and its trailing passes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accbpg_and_fw_amd as acc
m, n = 8192, 16384
V = torch.randn(m, n, dtype=torch.float64, device="cuda")
f = acc.DOptimalObj(V)
x = torch.rand(n, dtype=torch.float64, device="cuda") + 0.5
x /= x.sum()
for _ in range(5):
    fx, g = f.func_grad(x, 2)
print(fx)
