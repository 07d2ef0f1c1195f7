"""KG encoder (HGT over the whole graph) forward + backward on the training path: wall time per pass and the edge-attention
kernels' share (madrigal/models/models.py:71-96).  MDG_AB_LIB=<lib> times another build of the library.

    python scripts/kg_attention_bench.py [--precision bf16] [--reps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import _lib
if os.environ.get("MDG_AB_LIB"):
    _lib.LIB_PATH = os.environ["MDG_AB_LIB"]
from madrigal_amd import configs, data as D, models as M
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
M.set_precision(a.precision)
batch, bkg = D.make_batch(4096, 0, kg_nodes=130000, kg_edges=8000000)      # (the KG holds the batch's drugs: the bench shape)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], 8).cuda().train()
kg = bkg["data"].to("cuda")
enc = model.encoder
def kg_step():
    model.zero_grad(set_to_none=True)
    out = enc.kg_encoder(kg.x_dict, kg.edge_index_dict, only_types=("drug",))["drug"]
    out.sum().backward()
for _ in range(2):
    kg_step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(a.reps):
    e0.record(); kg_step(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"KG encoder fwd + bwd ({a.precision}, lib {os.path.basename(_lib.LIB_PATH)}): {best:.2f} ms")
