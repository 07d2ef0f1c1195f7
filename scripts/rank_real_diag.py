"""What the bench's own score tensor looks like to a sort: duplicate drug embeddings, tie groups, bucket balance.  python scripts/rank_real_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M, ops
from madrigal_amd.pipeline import generate_embeddings, score_all_pairs
L = 8
N = 4096
batch, bkg = D.make_batch(N, 0, kg_nodes=130_000, kg_edges=8_000_000)
model = configs.build_model("twosides321", bkg["data"], L).cuda().eval()
with torch.no_grad():
    model.decoder.parametrizations.weight.original.copy_(torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000)) / 128 ** 0.5)
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
filler = torch.randn(N, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
with torch.no_grad(), M.precision("bf16x3"):
    z = generate_embeddings(model, b, kgc, kg_filler=filler)
    s = score_all_pairs(model, z)
uz = torch.unique(z, dim=0)
print("distinct z rows:", uz.shape[0], "of", N)
il = torch.tril_indices(N, N, -1).cuda()
for l in range(3):
    v = s[l][il[0], il[1]]
    u, c = torch.unique(v, return_counts=True)
    cs = torch.sort(c, descending=True).values
    print(f"outcome {l}: {v.numel()} keys, {u.numel()} distinct, largest tie groups {cs[:8].tolist()}, keys in groups > 128: {int(c[c > 128].sum())}")
    q = torch.quantile(v[::97].float(), torch.tensor([0.001, 0.01, 0.25, 0.5, 0.75, 0.99, 0.999], device="cuda"))
    print("   quantiles", [round(float(x), 4) for x in q])
flags = []
out = ops.rank_normalize(s, fallback_flags=flags)
print("flags", torch.cat(flags).tolist())
