"""Rank normalisation on the score tensor the bench's own model produces (not randn): time per outcome and how many outcomes the MSD
fast path hands back to the LSD sort, with the reasons.    python scripts/rank_real_scores.py [outcomes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M, ops
from madrigal_amd.pipeline import generate_embeddings, score_all_pairs
L = int(sys.argv[1]) if len(sys.argv) > 1 else 128
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
print("z: mean |z| per dim", float(z.abs().mean()), " std over drugs (mean over dims)", float(z.std(0).mean()))
v = s[0][torch.tril_indices(N, N, -1).unbind()[0].cuda(), torch.tril_indices(N, N, -1).unbind()[1].cuda()]
print("outcome 0 scores: min %.4g max %.4g mean %.4g std %.4g" % (float(v.min()), float(v.max()), float(v.mean()), float(v.std())))
out = ops.empty_scores(L, N, N, s.device)
ops.rank_normalize(s[:4], out=out[:4]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    flags = []
    e0.record(); ops.rank_normalize(s, out=out, fallback_flags=flags); e1.record(); torch.cuda.synchronize()
    f = torch.cat(flags)
    print(f"{e0.elapsed_time(e1) / L * 1e3:.1f} us per outcome; handed back: {int((f != 0).sum())} of {L}; reasons {sorted(set(f.tolist()))}")
os.environ["MDG_RANKS_MSD"] = "0"
from madrigal_amd._lib import lib
lib().mdg_tuning_reload()
ref = ops.empty_scores(L, N, N, s.device)
e0.record(); ops.rank_normalize(s, out=ref); e1.record(); torch.cuda.synchronize()
print(f"LSD: {e0.elapsed_time(e1) / L * 1e3:.1f} us per outcome; identical: {bool(torch.equal(ref, out))}")
