"""Synchronising calls inside one contrastive-pretraining iteration (view draw + upload + PretrainStep.step)."""
import os, sys, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrigal_amd import configs, data as D, masks as MK, models as M
from madrigal_amd.optim import AdamW
from madrigal_amd.simclr import SimCLR_NovelDDI
from madrigal_amd.train import PretrainStep

M.set_precision("bf16x3")
B = 2048
avail = D.make_masks(B, 0)
avail[:, 1] = torch.where(avail[:, 1:].all(dim=1), torch.zeros(B, dtype=torch.bool), avail[:, 1])
batch, bkg = D.make_batch(B, 0, kg_nodes=130_000, kg_edges=8_000_000, masks=avail)
torch.manual_seed(0); np.random.seed(0)
enc = configs.build_model("twosides321", bkg["data"], 8).encoder
model = SimCLR_NovelDDI(enc, dim=128, mlp_dim=512, T=0.1, raw_encoder_output=True).cuda().train()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
draw = MK.StrCenterUniSampler(MK.get_pretrain_masks(list(range(B)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2))
step = PretrainStep(model, AdamW(model.parameters(), lr=1e-5, weight_decay=1e-2))
data = (b["strs"], kgc, b["cv"], b["tx"])
def one():
    m1, m2 = draw(range(B))
    if len(sys.argv) > 1 and sys.argv[1] == "device":
        return step.step(b["drugs"], m1.cuda(), m2.cuda(), None, data)
    return step.step(batch["drugs"], m1, m2, None, data)
for _ in range(3): one()
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f for f in traceback.extract_stack() if "madrigal_amd" in f.filename or "sync_points" in f.filename]
        seen[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-4:])] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
one()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in one iteration:", sum(seen.values()))
for k, v in seen.most_common():
    print(f"  {v:3d}  {k}")
