"""Consecutive scoring jobs pipelined over two streams: encode+fuse of job i+1 (stream A) beside the head of job i (stream B), against
the plain back-to-back order.  Same model / batch as bench.py's headline (4096 drugs, 896 outcomes, bf16x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M, ops
from madrigal_amd.pipeline import generate_embeddings

N, L, K = 4096, 896, 12
M.set_precision("bf16x3")
batch, bkg = D.make_batch(N, 0, kg_nodes=130000, kg_edges=8000000)
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
model = configs.build_model("twosides321", bkg["data"], L).to(dev).eval()
batch = D.batch_to(batch, dev)
bkg = {"data": bkg["data"].to(dev), "drug_index_map": bkg["drug_index_map"].to(dev)}
filler = torch.randn(N, 128, device=dev)
out = ops.empty_scores(L, N, N, dev)


@torch.no_grad()
def plain(k):
    for _ in range(k):
        z = generate_embeddings(model, batch, bkg, kg_filler=filler)
        model.decoder(z, z, (0, L), out=out)


@torch.no_grad()
def piped(k):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()
    sa.wait_stream(main)
    sb.wait_stream(main)
    zs, enc_done, head_done = [None, None], [None, None], [None, None]
    for i in range(k):
        b = i & 1
        with torch.cuda.stream(sa):
            if head_done[b] is not None:
                sa.wait_event(head_done[b])           # z buffer b is free once the head of job i-2 has read it
            z = generate_embeddings(model, batch, bkg, kg_filler=filler)
            zs[b] = z
            enc_done[b] = torch.cuda.Event()
            enc_done[b].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(enc_done[b])
            model.decoder(zs[b], zs[b], (0, L), out=out)
            head_done[b] = torch.cuda.Event()
            head_done[b].record(sb)
    main.wait_stream(sa)
    main.wait_stream(sb)


for name, fn in (("plain", plain), ("piped", piped), ("plain", plain), ("piped", piped)):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{name}: {dt * 1e3:.2f} ms per job = {L * N * N / dt:.3e} scores/s", flush=True)
