"""Finetune step with a NEW set of labelled triples every step (what an epoch looks like: the triple plan -- sorts by label,
by (label, head) pair and by drug -- is rebuilt per batch) against the cached-plan steady state train_bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data, models as M
from madrigal_amd.optim import create_optimizer
from madrigal_amd.train import FinetuneStep
M.set_precision("bf16")
batch, bkg = data.make_batch(4096, seed=0, kg_nodes=130_000, kg_edges=8_000_000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], n_outcomes=896).cuda()
b = data.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
sets = [tuple(t.cuda() for t in data.make_labelled_triples(4096, 896, 1_000_000, s)) for s in range(4)]
filler = torch.randn(4096, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
          wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
for mode in ("same triples", "fresh triples"):
    for i in range(3):
        fs.step(b, b, b["masks"], b["masks"], kgc, *sets[0], kg_filler=filler)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 8
    for i in range(n):
        fs.step(b, b, b["masks"], b["masks"], kgc, *(sets[0] if mode == "same triples" else sets[i % 4]), kg_filler=filler)
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t) / n * 1e3:.1f} ms per step")
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(4):
    fs.plan(*sets[i][:3], 4096, 4096)
torch.cuda.synchronize()
print(f"triple plan alone: {(time.perf_counter() - t) / 4 * 1e3:.1f} ms")
