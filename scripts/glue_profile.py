"""Which call sites of the finetune step launch torch's own small kernels (fill / add / copy / cat / index)?  torch profiler with
Python stacks; prints, per aten op, the innermost madrigal_amd frames and their counts per step."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from madrigal_amd import configs, data as D, models as M
from madrigal_amd.optim import create_optimizer
from madrigal_amd.train import FinetuneStep
M.set_precision("bf16")
batch, bkg = D.make_batch(4096, 0, kg_nodes=130000, kg_edges=8000000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], 896).cuda()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
lab, hd, tl, y = (t.cuda() for t in D.make_labelled_triples(4096, 896, 1000000, 0))
filler = torch.randn(4096, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4, wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
for _ in range(2):
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::cat", "aten::index_select", "aten::index_add_", "aten::index", "aten::index_put_", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous")
agg = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if "madrigal_amd" in f or "autograd" in f.lower()]
        key = " <- ".join(f.split("/")[-1][:70] for f in frames[:2]) or "(engine / no python frame)"
        agg[ev.name][key] += 1
for op in want:
    tot = sum(agg[op].values())
    if tot:
        print(f"== {op}: {tot}")
        for k, c in agg[op].most_common(8):
            print(f"   {c:5d}  {k}")
