import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops, autograd as ag
x = torch.randn(64, 128, device="cuda"); w = torch.randn(128, 128, device="cuda"); b = torch.randn(128, device="cuda")
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    print(f"{name:34s} {dt * 1e6:7.1f} us/call (host issue)")
bench("ops.linear cache_weight=False", lambda: ops.linear(x, w, b, cache_weight=False))
bench("ops.linear cached weight", lambda: ops.linear(x, w, b))
bench("ops.transpose", lambda: ops.transpose(w))
bench("ops.grad_weight +bias", lambda: ops.grad_weight(x, x, "bf16x3", want_bias=True))
bench("ops.activation_bwd", lambda: ops.activation_bwd(x, x, "gelu"))
bench("ops.layernorm", lambda: ops.layernorm(x, b, b))
bench("ops.colsum", lambda: ops.colsum(x))
bench("torch.empty", lambda: torch.empty(64, 128, device="cuda"))
bench("torch add (reference op)", lambda: x + x)
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
def fb():
    y = ag.linear(xg, wg, bg, "gelu")
    y.backward(x[:, :128])
bench("ag.linear fwd+bwd (gelu)", fb, 500)
lin = torch.nn.Linear(128, 128).cuda()
def tfb():
    y = torch.nn.functional.gelu(lin(xg))
    y.backward(x)
bench("torch linear+gelu fwd+bwd", tfb, 500)
