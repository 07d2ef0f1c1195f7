"""Does the inference encode survive hipGraph capture (torch.cuda.CUDAGraph), and what does replay buy?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M
for N, kgn, kge in ((4096, 130000, 8000000), (256, 20000, 400000)):
    batch, bkg = D.make_batch(N, 0, kg_nodes=kgn, kg_edges=kge)
    torch.manual_seed(0)
    model = configs.build_model("twosides321", bkg["data"], 64).cuda().eval()
    b = D.batch_to(batch, "cuda")
    kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
    filler = torch.randn(N, 128, device="cuda")
    enc = model.encoder
    def run():
        return enc(b["drugs"], b["masks"], b["strs"], kgc, b["cv"], b["tx"], kg_filler=filler)
    with torch.no_grad():
        for _ in range(3):
            z_ref = run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t) / 10
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            z = run()
        g.replay()
        torch.cuda.synchronize()
        ok = torch.equal(z, z_ref)
        t = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t) / 10
    print(f"N={N}: eager {eager * 1e3:.2f} ms, graph replay {graphed * 1e3:.2f} ms, identical={ok}", flush=True)
