"""numpy model of the MSD rank path's bucket function (csrc/ranks.hip: msd_level1 / msd_table / msd_bucket_of): bucket sizes for a
score vector in triangle order.  Used to tune the sampling pattern; `python scripts/rank_bucket_sim.py real` runs it on the bench's own
score tensor (GPU box), without arguments on synthetic shapes (CPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def keys_of(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def sim(vals, NC=4096, QLG=12, CH=4096, D=256, F=4, name="", chunk=64, verbose=True):
    k = keys_of(vals).astype(np.int64)
    M = k.size
    n_chunks = (M + chunk - 1) // chunk
    chs = CH * 64 // chunk
    S = (n_chunks + chs - 1) // chs
    n_groups = (n_chunks + S - 1) // S
    g = np.arange(n_groups, dtype=np.uint64)
    h = (g * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    c = np.minimum(g * S + ((h >> np.uint64(7)) % np.uint64(S)), n_chunks - 1).astype(np.int64)
    idx = (c[:, None] * chunk + np.arange(chunk)[None, :]).ravel()
    idx = idx[idx < M]
    smp = k[idx]
    ns = smp.size
    lo = int(smp.min()); hi = int(smp.max())
    s1 = 0
    while s1 < 23 and (hi >> s1) - (lo >> s1) >= 512:
        s1 += 1
    e0 = lo >> s1
    h1 = np.bincount((np.clip(smp, lo, hi) >> s1) - e0, minlength=512)
    lgb = np.zeros(512, dtype=np.int64)
    big = h1 > D
    lgb[big] = np.ceil(np.log2(np.ceil(h1[big] / D))).astype(np.int64)
    lgb = np.minimum(lgb, s1)
    base = np.concatenate([[0], np.cumsum(1 << lgb)[:-1]])

    def coarse(kk):
        kc = np.clip(kk, lo, hi); e = (kc >> s1) - e0; se = s1 - lgb[e]; low1 = kc & ((1 << s1) - 1)
        return base[e] + (low1 >> se), low1 & ((1 << se) - 1), se
    cs, _, _ = coarse(smp)
    h2 = np.bincount(cs, minlength=NC)
    nbt = (M + (1 << QLG) - 1) >> QLG
    unit = min(15, max(1, ns // nbt // 8))
    e_of_bin = np.searchsorted(base, np.arange(NC), side='right') - 1
    se_bin = s1 - lgb[e_of_bin]
    cum = np.concatenate([[0], np.cumsum(h2)[:-1]])
    lg = np.zeros(NC, dtype=np.int64)
    bb = h2 > unit
    parts = (h2 + unit - 1) // unit
    lg[bb] = np.ceil(np.log2(parts[bb])).astype(np.int64)
    lg = np.minimum(lg, se_bin)
    perF = np.minimum((h2 * F) >> lg, 63)
    mul = (nbt << 32) // (F * ns)
    cb, low2, se = coarse(k)
    sub = low2 >> (se - lg[cb])
    x = cum[cb] * F + sub * perF[cb]
    b = np.minimum((x * mul) >> 32, nbt - 1)
    cnt = np.bincount(b, minlength=nbt)
    if verbose:
        print(f"{name:12s} chunk {chunk:3d} M {M} ns {ns} s1 {s1} bins {int((1 << lgb).sum())} nbt {nbt} max bucket {cnt.max()} mean {cnt.mean():.0f} "
              f"ratio {cnt.max() / max(1, cnt.mean()):.2f}  buckets > 6144: {(cnt > 6144).sum()}", flush=True)
    sim.last = dict(h1=h1, h2=h2, lgb=lgb, base=base, lo=lo, hi=hi, s1=s1, ent=((cum * F) << 11) | (perF << 5) | lg, mul=mul, ns=ns)
    return cnt


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "real":
        import torch
        from madrigal_amd import configs, data as D_, models as M_
        from madrigal_amd.pipeline import generate_embeddings, score_all_pairs
        L, N = 8, 4096
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
        il = np.tril_indices(N, -1)
        zz = z.cpu().numpy()
        print("z row norms: mean %.3f std %.3f; correlation of adjacent rows' z: %.3f" % (np.linalg.norm(zz, axis=1).mean(), np.linalg.norm(zz, axis=1).std(),
              np.mean([np.corrcoef(zz[i], zz[i + 1])[0, 1] for i in range(0, 4000, 40)])))
        for l in (0, 1, 3):
            v = s[l].cpu().numpy()[il]
            for chunk in (64, 16, 4, 1):
                sim(v, name=f"real[{l}]", chunk=chunk)
        sys.exit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(0)
    il = np.tril_indices(N, -1)
    x = rng.standard_normal((N, N)).astype(np.float32)
    a = rng.standard_normal(N)[:, None]
    for chunk in (64, 16, 1):
        sim(x[il], name="randn", chunk=chunk)
        sim((3 + a + a.T + 0.3 * x)[il], name="rowstruct", chunk=chunk)
        sim((3 + a + a.T + 0.03 * x)[il], name="rowstruct.03", chunk=chunk)


def fine_bin_report(vals, N, NF=16384, cap=12288, QLG=13):
    """Largest fine-bin occupancy of the bucket sort's composite (u(key), position) bins, per bucket: numpy model of MsdFine in
    csrc/ranks.hip (u = the key, or the fixed-point score where the bucket holds both signs or is massed at its large-magnitude end)."""
    k = keys_of(vals).astype(np.int64)
    il = np.tril_indices(N, -1)
    q26 = (il[0].astype(np.int64) << 13) | il[1].astype(np.int64)
    cnt = sim(vals, verbose=False, QLG=QLG)
    order = np.argsort(k, kind="stable")                 # (equi-depth buckets of the model's sizes, in key order)
    edges = np.concatenate([[0], np.cumsum(cnt)])
    v32 = np.ascontiguousarray(vals, dtype=np.float32)
    worst = []
    for b in range(cnt.size):
        idx = order[edges[b]:edges[b + 1]]
        if idx.size == 0 or idx.size > cap:
            continue
        kk = k[idx]; kmin, kmax = int(kk.min()), int(kk.max())
        pos = (kk.mean() - kmin) / (kmax - kmin + 1.0)
        both = kmin < 0x80000000 <= kmax
        fixed = both or (kmin >= 0x80000000 and pos > 0.6) or (kmax < 0x80000000 and pos < 0.4)
        if fixed:
            sc = v32[idx].astype(np.float64)
            m, mn = np.abs(sc).max(), np.abs(sc).min()
            e = 29 - int(np.floor(np.log2(m))) if m > 0 else 0
            if not both and mn > 0:
                e = min(e, 23 - int(np.floor(np.log2(mn))))
            u = np.rint(np.ldexp(sc, e)).astype(np.int64)
        else:
            u = kk
        u = u - u.min()
        span = (int(u.max()) << 26) | 0x3FFFFFF
        sh = max(0, span.bit_length() - int(np.log2(NF)))
        f = ((u << 26) | q26[idx]) >> sh
        c = np.bincount(f, minlength=NF)
        worst.append((int(c.max()), b, idx.size, kmax - kmin, sh, bool(fixed), round(float(pos), 3)))
    worst.sort(reverse=True)
    return worst[:4]
