"""Compare the MSD rank path's device tables and bucket totals with the numpy model (scripts/rank_bucket_sim.py) on one outcome of the
bench's own score tensor.  python scripts/rank_msd_debug.py [outcome]"""
import os, sys
os.environ["MDG_RANKS_GROUP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from madrigal_amd import configs, data as D_, models as M_, ops
from madrigal_amd.pipeline import generate_embeddings, score_all_pairs
from rank_bucket_sim import sim
L, N = 8, 4096
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
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
G = int(os.environ.get("DBG_G", "8"))
os.environ["MDG_RANKS_GROUP"] = str(G)
from madrigal_amd._lib import lib
lib().mdg_tuning_reload()
sub = s[:G]
print("layout: shape", tuple(sub.shape), "strides", sub.stride())
flags = []
ops.rank_normalize(sub, fallback_flags=flags)
torch.cuda.synchronize()
print("flags", flags[0].tolist())
ws = max(ops._ws_cache.values(), key=lambda t: t.numel())
a256 = lambda x: (x + 255) // 256 * 256
M = N * (N - 1) // 2
n_tiles = ((N + 127) // 128) * ((N + 127) // 128 + 1) // 2      # a count / partition tile is a 128 x 128 output block
nbt = (M + 8191) // 8192                             # MSD_QLG = 13
nbs = (nbt + 255) // 256 * 256
off = 2 * a256(4 * (G + 1))                          # flags | list of handed-back outcomes
mm = ws[off:off + 8 * G].view(torch.int32).cpu().numpy().view(np.uint32); off += a256(8 * G)
hist = ws[off:off + G * (512 + 4096) * 4].view(torch.int32).cpu().numpy().view(np.uint32); off += a256(G * (512 + 4096) * 4)
tabs = ws[off:off + G * 4624 * 4].view(torch.int32).cpu().numpy().view(np.uint32).reshape(G, 4624); off += a256(G * 4624 * 4)
off += a256(G * M * 8) + a256(G * M * 4)           # pairs by bucket | rank words
off += a256(G * n_tiles * nbs * 2)
off += a256(G * n_tiles * nbs * 4)
totals_all = ws[off:off + G * nbs * 4].view(torch.int32).cpu().numpy().view(np.uint32).reshape(G, nbs)
il = np.tril_indices(N, -1)
for which in range(G if os.environ.get("DBG_MODEL", "1") != "0" else 0):
    v = sub[which].cpu().numpy()[:, :N][il]
    cnt = sim(v, name=f"model[{which}]")
    m = sim.last
    tab = tabs[which]
    totals = totals_all[which]
    h1 = hist[:G * 512].reshape(G, 512)[which]
    h2 = hist[G * 512:].reshape(G, 4096)[which]
    print("  device lo/hi", hex(mm[2 * which]), hex(mm[2 * which + 1]), "model", hex(m["lo"]), hex(m["hi"]), "hdr", [hex(x) for x in tab[:6]], "model s1", m["s1"], "mul", m["mul"])
    print("  hist1 equal:", np.array_equal(h1, m["h1"][:512]), " hist2 equal:", np.array_equal(h2, m["h2"]))
    t1 = tab[16:16 + 512]
    print("  level 1 equal:", np.array_equal(t1 & 0xFFFF, m["base"]) and np.array_equal(t1 >> 16, m["lgb"]), " level 2 equal:", np.array_equal(tab[16 + 512:], m["ent"].astype(np.uint32)))
    print("  device totals: sum", int(totals.sum()), "max", int(totals.max()), "at", int(totals.argmax()), " model max", int(cnt.max()), "at", int(cnt.argmax()),
          " equal:", np.array_equal(totals[:nbt], cnt))
    d = np.nonzero(totals[:nbt] != cnt)[0]
    print("     differing buckets:", d.size, d[:10], totals[d[:10]], cnt[d[:10]])
if os.environ.get("DBG_STAMPS"):
    # phase stamps of the persistent kernels (built with MDG_EXTRA_HIPCC_FLAGS=-DMDG_RANK_STAMPS): in the big-bucket scratch
    off += 2 * a256(G * nbs * 4)                      # totals, bases
    off += G * 63 * 65536 * 8 - 2 * 8 * 4096 * 8       # the tail of the scratch
    st = ws[off:off + 2 * 8 * 4096 * 8].view(torch.int64).cpu().numpy().reshape(2, 4096, 8)
    wgs = max(1, 256 // G)
    for name, blk, phases in (("partition (msd_partition_pipe_kernel)", st[0], ["-", "keys + prefetch issue", "slots of this tile | pairs of the previous tile out", "barrier", "scan (2 barriers)", "place + offsets", "end barrier", "-"]),
                              ("bucket sort", st[1], ["-", "init", "atomics", "barrier", "scan (2 barriers)", "bounds+place", "barrier+zero+probes+stores", "next stats+end barrier"])):
        a = blk[:G * wgs].astype(np.float64)
        a = a[a.sum(1) > 0]
        tot = a.sum(1)
        print(f"{name}: {a.shape[0]} workgroups, mean cycles per workgroup {tot.mean():.0f} (= {tot.mean() / 2.4e3:.1f} us at 2.4 GHz; clock units are the s_memtime counter's)")
        for i, ph in enumerate(phases):
            if i: print(f"   {ph:45s} {a[:, i].mean():10.0f}  {100 * a[:, i].mean() / tot.mean():5.1f} %")
