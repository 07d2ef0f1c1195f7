import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M
batch, bkg = D.make_batch(4096, 0, kg_nodes=130000, kg_edges=8000000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], 8).cuda().train()
kg = bkg["data"].to("cuda")
enc = model.encoder
b = D.batch_to(batch, "cuda")
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - s)
    return best * 1e3
def kg_step():
    model.zero_grad(set_to_none=True)
    out = enc.kg_encoder(kg.x_dict, kg.edge_index_dict, only_types=("drug",))["drug"]
    out.sum().backward()
def kg_fwd():
    with torch.no_grad():
        enc.kg_encoder(kg.x_dict, kg.edge_index_dict, only_types=("drug",))
def gin_step():
    model.zero_grad(set_to_none=True)
    enc.str_encoder(b["strs"], b["strs"].node_feature.float())["graph_feature"].sum().backward()
def tx_step():
    model.zero_grad(set_to_none=True)
    enc._encode_tx(b["tx"], 4096, "cuda").sum().backward()
print("kg fwd (train path, no grad) ms", t(kg_fwd))
print("kg fwd+bwd ms", t(kg_step))
print("gin fwd+bwd ms", t(gin_step))
print("tx fwd+bwd ms", t(tx_step))
