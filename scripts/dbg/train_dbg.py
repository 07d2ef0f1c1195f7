import sys, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from madrigal_amd import data as D, models as M, ops, autograd as ag
from test_train_gpu import _small_model
case = ("twosides321", "transformer_uni_proj", 2, "sinusoidal", 8, 256, 1024, 2, True, "x-attn", False, False)
n, L, seed = 128, 16, 5
model, p, batch, bkg, masks = _small_model(M, case, n, L, seed, default_init=True)
model = model.cuda().train()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
lab, hd, tl, y = (t.cuda() for t in D.make_labelled_triples(n, L, 3000, seed))
filler = torch.randn(n, 128, generator=torch.Generator().manual_seed(1)).cuda()
plan = ops.triple_plan(lab, hd, tl, L, n, n)
def loss_fn():
    torch.manual_seed(77)
    s = model.score_triples(b, b, b["masks"], b["masks"], kgc, plan, kg_filler=filler)
    return ag.bce_with_sigmoid(s, y), s
loss, s = loss_fn()
print("loss0", float(loss.detach()), "score abs max", float(s.detach().abs().max()), "score std", float(s.detach().std()))
loss.backward()
groups = {}
for k, q in model.named_parameters():
    g = "none" if q.grad is None else k.split(".")[1] if k.startswith("encoder.") else k.split(".")[0]
    groups.setdefault(g, []).append((k, q))
for g, items in groups.items():
    if g == "none":
        print("no grad:", [k for k, _ in items][:8]); continue
    gn = sum(float((q.grad ** 2).sum()) for _, q in items) ** 0.5
    for eps in (1e-3 / max(gn, 1e-12), 1e-2 / max(gn, 1e-12)):
        with torch.no_grad():
            for _, q in items: q.sub_(eps * q.grad)
            l1, _ = loss_fn()
            for _, q in items: q.add_(eps * q.grad)
        pred = -eps * gn * gn
        print(f"{g:22s} |g|={gn:9.3e} eps={eps:9.3e} dL={float(l1)-float(loss.detach()):+.4e} predicted={pred:+.4e}")
