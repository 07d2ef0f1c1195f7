import copy, torch, sys
sys.path.insert(0, ".")
from madrigal_amd import autograd as ag, ops
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(300, 128, generator=g))
dy = torch.randn(300, 128, generator=g)
w = torch.ones(128); b = torch.zeros(128)
for name, xx in (("relu-input", x), ("dense-input", x + 0.1)):
    xr, wr, br = (v.double().requires_grad_(True) for v in (xx, w, b))
    torch.nn.functional.layer_norm(xr, (128,), wr, br, 1e-5).backward(dy.double())
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), xx.cuda(), w.cuda(), 1e-5)
    print(name, float((dx.cpu().double() - xr.grad).abs().max()), float(xr.grad.abs().max()),
          float((dg.cpu().double() - wr.grad).abs().max()), float((db.cpu().double() - br.grad).abs().max()))
    bad = (dx.cpu().double() - xr.grad).abs().max(1).values
    print(" worst rows", bad.topk(5))
