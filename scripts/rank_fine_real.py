import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import numpy as np, torch
from madrigal_amd import configs, data as D_, models as M_, ops
from madrigal_amd.pipeline import generate_embeddings, score_all_pairs
from rank_bucket_sim import fine_bin_report, keys_of
L, N = 896, 4096
batch, bkg = D_.make_batch(N, 0, kg_nodes=130_000, kg_edges=8_000_000)
model = configs.build_model("twosides321", bkg["data"], L).cuda().eval()
with torch.no_grad():
    model.decoder.parametrizations.weight.original.copy_(torch.randn(L, 128, 128, generator=torch.Generator().manual_seed(1000)) / 128 ** 0.5)
b = D_.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
filler = torch.randn(N, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
with torch.no_grad(), M_.precision("bf16x3"):
    z = generate_embeddings(model, b, kgc, kg_filler=filler)
    s = score_all_pairs(model, z)
flags = []
ops.rank_normalize(s, fallback_flags=flags)

il = np.tril_indices(N, -1)
fl = torch.cat(flags).tolist()
print("flagged outcomes:", [i for i, f in enumerate(fl) if f])
for l in [i for i, f in enumerate(fl) if f][:4] + [0]:
    v = s[l].cpu().numpy()[:, :N][il]
    w = fine_bin_report(v, N)
    print(l, "flag", fl[l], "worst fine bins (count, bucket, n, key range, shift, fixed-point, mean key position):", w, flush=True)
