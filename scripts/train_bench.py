"""Finetune-step timing at the BASELINE configs[1] shape: 4096 drugs, 896 outcomes, the TWOSIDES fusion model over a
130k-node / 8M-edge KG, T labelled triples.  Prints ms per step and a coarse phase split (HIP events)."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from madrigal_amd import configs, data, models as M, ops, autograd as ag  # noqa: E402
from madrigal_amd.optim import create_optimizer  # noqa: E402
from madrigal_amd.train import FinetuneStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drugs", type=int, default=4096)
ap.add_argument("--outcomes", type=int, default=896)
ap.add_argument("--triples", type=int, default=1_000_000, help="positives; x6 labelled triples (2 negatives each, both directions)")
ap.add_argument("--kg-nodes", type=int, default=130_000)
ap.add_argument("--kg-edges", type=int, default=8_000_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--config", default="twosides321")
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--phases", action="store_true")
a = ap.parse_args()

M.set_precision(a.precision)
batch, bkg = data.make_batch(a.drugs, seed=0, kg_nodes=a.kg_nodes, kg_edges=a.kg_edges)
torch.manual_seed(0)
model = configs.build_model(a.config, bkg["data"], n_outcomes=a.outcomes).cuda()
b = data.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
lab, hd, tl, y = (t.cuda() for t in data.make_labelled_triples(a.drugs, a.outcomes, a.triples, 0))
filler = torch.randn(a.drugs, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
          wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
losses = []
for i in range(a.warmup + a.steps):
    if i == a.warmup:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    losses.append(fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler))
host_issue = (time.perf_counter() - t0) / a.steps          # the host is done queueing here; the GPU may still be working
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
out = {"ms_per_step": dt * 1e3, "host_issue_ms_per_step": host_issue * 1e3, "steps_per_sec": 1 / dt, "loss": [float(x) for x in losses], "drugs": a.drugs, "outcomes": a.outcomes,
       "triples": int(lab.numel()), "precision": a.precision, "max_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
if a.phases:
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    model.train()
    fs.optimizer.zero_grad(set_to_none=True)
    plan = fs.plan(lab, hd, tl, a.drugs, a.drugs)
    ev[0].record()
    zh, zt = model.embed(b, b, b["masks"], b["masks"], kgc, kg_filler=filler)
    ev[1].record()
    s = model.decoder.score_triples(zh, zt, plan)
    loss = ag.bce_with_sigmoid(s, y[plan["perm"]])
    ev[2].record()
    loss.backward()
    ev[3].record()
    fs.optimizer.step()
    ev[4].record()
    torch.cuda.synchronize()
    out["phases_ms"] = dict(zip(["encode_fwd_x2", "head_fwd+loss", "backward", "adamw"], [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]))
print(json.dumps(out))
