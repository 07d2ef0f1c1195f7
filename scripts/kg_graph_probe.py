"""KG encoder training pass: eager launches against the captured graphs (forward + backward of the drug rows)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data, models as M
M.set_precision("bf16")
batch, bkg = data.make_batch(4096, seed=0, kg_nodes=130_000, kg_edges=8_000_000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], n_outcomes=8).cuda().train()
enc = model.encoder
kg = bkg["data"].to("cuda")
dy = None
def run(graphed):
    global dy
    f = enc._kg_graphed(kg, torch.device("cuda")) if graphed else None
    if graphed:
        print("graphed runner:", type(f).__name__ if f is not None else None)
    def once():
        global dy
        out = f(kg.x_dict["drug"]) if f is not None else enc.kg_encoder(kg.x_dict, kg.edge_index_dict, only_types=("drug",))["drug"]
        if dy is None:
            dy = torch.randn_like(out)
        out.backward(dy)
        return out
    for _ in range(3): once()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = once()
    host = (time.perf_counter() - t) / 10
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t) / 10
    g = {k: p.grad.clone() for k, p in enc.kg_encoder.named_parameters() if p.grad is not None}
    enc.kg_encoder.zero_grad()
    print(f"{'graphed' if graphed else 'eager  '}: host {host*1e3:.2f} ms, total {tot*1e3:.2f} ms per fwd+bwd")
    return out.detach().clone(), g
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    o1, g1 = run(False)
    o2, g2 = run(True)
    for x in w:
        if "capture" in str(x.message): print("WARNING:", str(x.message)[:400])
print("outputs equal:", torch.equal(o1, o2), "grad keys equal:", set(g1) == set(g2))
# one backward accumulates: compare single-step grads
enc.kg_encoder.zero_grad()
