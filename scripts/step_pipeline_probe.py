"""Where the finetune step is bound by the host's launch rate and where by the GPU.

At marks inside a steady-state step (start, encoders + fusion queued, head + loss queued, gradient of the KG rows ready = the KG
encoder's backward begins, backward queued, AdamW queued) an event is recorded and the host clock read.  ``lag`` = when the GPU
reached the mark minus when the host queued it: a small lag means the GPU was waiting for the host there (launch-bound section
before the mark), a large one that the host runs ahead (GPU-bound).

    python scripts/step_pipeline_probe.py [--precision bf16] [--steps 12]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data, models as M, ops, autograd as ag
from madrigal_amd.optim import create_optimizer
from madrigal_amd.train import FinetuneStep

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=12)
a = ap.parse_args()
M.set_precision(a.precision)
batch, bkg = data.make_batch(4096, seed=0, kg_nodes=130_000, kg_edges=8_000_000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], n_outcomes=896).cuda()
b = data.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
lab, hd, tl, y = (t.cuda() for t in data.make_labelled_triples(4096, 896, 1_000_000, 0))
filler = torch.randn(4096, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
          wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
marks = []            # (step, name, host time, event)
cur = [0]
def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    marks.append((cur[0], name, time.perf_counter(), ev))
orig_embed, orig_score, orig_gather = model.embed, model.decoder.score_triples, ag.gather_rows_or
def embed(*x, **k):
    r = orig_embed(*x, **k); mark("encoders+fusion fwd queued"); return r
def score(*x, **k):
    r = orig_score(*x, **k); mark("head fwd queued"); return r
def gather(*x, **k):
    r = orig_gather(*x, **k)
    if r.requires_grad:
        r.register_hook(lambda g: mark("KG rows' gradient ready (KG backward begins)"))
    return r
model.embed, model.decoder.score_triples, ag.gather_rows_or = embed, score, gather
orig_backward = torch.Tensor.backward
def step():
    mark("step start")
    model.train()
    fs.optimizer.zero_grad(set_to_none=True)
    loss = fs.accumulate(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
    mark("backward queued")
    fs.apply()
    mark("AdamW queued")
for _ in range(4):
    step()
torch.cuda.synchronize()
marks.clear()
ref = torch.cuda.Event(enable_timing=True)
ref.record(); torch.cuda.synchronize()
t_ref = time.perf_counter()
for i in range(a.steps):
    cur[0] = i
    step()
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f"{a.steps} steps: {(t_end - t_ref) / a.steps * 1e3:.2f} ms per step ({a.precision})")
rows = [(s, n, (t - t_ref) * 1e3, ref.elapsed_time(e)) for s, n, t, e in marks]
for s in (a.steps - 3, a.steps - 2):
    base = [r for r in rows if r[0] == s][0][2]
    print(f"step {s}:   mark                                             host queued   GPU reached   lag (GPU - host), ms after the step's start on the host")
    for _, n, th, tg in [r for r in rows if r[0] == s]:
        print(f"   {n:52s} {th - base:9.2f}   {tg - base:9.2f}   {tg - th:9.2f}")
