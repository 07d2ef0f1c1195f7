import copy, torch, sys
sys.path.insert(0, ".")
from madrigal_amd import models as M
torch.manual_seed(0)
m = M.MLPEncoder(50, [128, 256, 64], 128, 0.0, "ln", "relu", "nd")
ref = copy.deepcopy(m).double().train()
m = m.cuda().train()
g = torch.Generator().manual_seed(11)
x = torch.randn(300, 50, generator=g)
dy = torch.randn(300, 128, generator=g)
for prec in ("f32", "bf16x3"):
    M.set_precision(prec)
    m.zero_grad(); ref.zero_grad()
    xg = x.cuda().requires_grad_(True); xr = x.double().requires_grad_(True)
    yg = m(xg); yr = ref.fc(xr)
    print(prec, "fwd", float((yg.cpu().double() - yr).abs().max() / yr.abs().max()))
    yg.backward(dy.cuda()); yr.backward(dy.double())
    for (n, pg), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        print("  ", n, float((pg.grad.cpu().double() - pr.grad).abs().max() / pr.grad.abs().max()), float(pr.grad.abs().max()))
    print("   dx", float((xg.grad.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max()))
