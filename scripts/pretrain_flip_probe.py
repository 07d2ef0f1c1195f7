"""Is a gradient outlier of test_simclr_raw_training_step_matches_oracle_autograd a ReLU flip (rounding-distance pre-activation) or an error?
Runs the test's body over several seeds and prints the three worst parameters of each; a flip shows as an isolated outlier on one seed
whose error concentrates in few ENTRIES of one tensor (printed: share of entries off by more than 1e-3 of the tensor's scale)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_pretrain_gpu as T
from madrigal_amd import data as D, models as M
from oracle import madrigal_oracle as O
from oracle.pipeline import oracle_simclr

for seed in [int(s) for s in (sys.argv[1:] or ["33", "34", "35", "36"])]:
    n, Tm = 72, 0.1
    avail, m1, m2 = T._views(n, seed)
    batch, bkg = D.make_batch(n, seed, kg_nodes=500, kg_edges=5000, masks=avail)
    hard = torch.rand(n, n, generator=torch.Generator().manual_seed(3)) < 0.04
    hard = (hard | hard.T) & ~torch.eye(n, dtype=torch.bool)
    torch.manual_seed(seed)
    model = T._no_dropout(T._build(M, bkg["data"], False, True, mlp_dim=256, T=Tm))
    p0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    filler = torch.zeros(max(int(batch["drugs"].max()) + 1, int(bkg["drug_index_map"].max()) + 1), 128)
    pr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p0.items()}
    with O.batch_statistics({}):
        ref = oracle_simclr(pr, batch, bkg, m1, m2, hard, Tm, filler, shared_predictor=False, use_tx_basal=True)
    ref["loss"].backward()
    model = model.cuda().train()
    b = D.batch_to(batch, "cuda")
    kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
    with M.precision("f32"):
        a1, a2, (lg, lb, loss) = model(b["drugs"], m1.cuda(), m2.cuda(), hard.cuda(), (b["strs"], kgc, b["cv"], b["tx"]))
        loss.backward()
    named = dict(model.named_parameters())
    gmax = max(float(v.grad.abs().max()) for v in pr.values() if torch.is_tensor(v) and v.grad is not None)
    rows = []
    for k, v in pr.items():
        if k not in named or not (torch.is_tensor(v) and v.requires_grad) or v.grad is None or not bool(v.grad.any()):
            continue
        a, r = named[k].grad.cpu().double(), v.grad.double()
        scale = max(float(r.abs().max()), 1e-2 * gmax)
        d = (a - r).abs() / scale
        rows.append((float(d.max()), k, float((d > 1e-3).double().mean()), d.numel()))
    rows.sort(reverse=True)
    print(f"seed {seed}: loss {float(loss):.6f} vs {float(ref['loss']):.6f}")
    for e, k, share, numel in rows[:4]:
        print(f"   {e:.2e}  {k}  entries off by > 1e-3 of scale: {share:.4f} of {numel}")
