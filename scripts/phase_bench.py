import sys, time, torch
sys.path.insert(0, '.')
from madrigal_amd import data as D, configs, ops, models as M
from madrigal_amd.pipeline import generate_embeddings
cfg = sys.argv[1] if len(sys.argv) > 1 else 'twosides321'
N, L = 4096, 896
t0 = time.time(); batch, bkg = D.make_batch(N, 0, kg_nodes=130000, kg_edges=8000000); print('gen', time.time()-t0, flush=True)
model = configs.build_model(cfg, bkg['data'], L).cuda().eval()
b = D.batch_to(batch, 'cuda'); kgc = {'data': bkg['data'].to('cuda'), 'drug_index_map': bkg['drug_index_map'].cuda()}
enc = model.encoder
filler = torch.randn(N, 128, device='cuda')
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        t=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t)
    return min(ts)*1e3
for prec in ('bf16x3','f32'):
  with torch.no_grad(), M.precision(prec):
    print(prec, 'str  ms', timeit(lambda: enc.str_encoder(b['strs'], b['strs'].node_feature.float())))
    print(prec, 'kg   ms', timeit(lambda: enc.kg_encoder(kgc['data'].x_dict, kgc['data'].edge_index_dict)))
    print(prec, 'cv   ms', timeit(lambda: enc.cv_encoder(b['cv'])))
    print(prec, 'tx   ms', timeit(lambda: enc._encode_tx(b['tx'], N, 'cuda')))
    print(prec, 'enc  ms', timeit(lambda: enc(b['drugs'], b['masks'], b['strs'], kgc, b['cv'], b['tx'], kg_filler=filler)))
    z = enc(b['drugs'], b['masks'], b['strs'], kgc, b['cv'], b['tx'], kg_filler=filler)
    print('z finite', bool(torch.isfinite(z).all()), float(z.abs().mean()))
    out = torch.empty(L, N, N, device='cuda')
    print(prec, 'head ms', timeit(lambda: model.decoder(z, z, out=out)), flush=True)
