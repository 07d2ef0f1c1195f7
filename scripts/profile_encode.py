import sys, time, torch
sys.path.insert(0, '.')
from madrigal_amd import data as D, configs, ops, models as M
cfg = sys.argv[1] if len(sys.argv) > 1 else 'twosides321'
N, L = 4096, 896
batch, bkg = D.make_batch(N, 0, kg_nodes=130000, kg_edges=8000000)
model = configs.build_model(cfg, bkg['data'], L).cuda().eval()
b = D.batch_to(batch, 'cuda'); kgc = {'data': bkg['data'].to('cuda'), 'drug_index_map': bkg['drug_index_map'].cuda()}
enc = model.encoder
filler = torch.randn(N, 128, device='cuda')
with torch.no_grad():
    for _ in range(2):
        z = enc(b['drugs'], b['masks'], b['strs'], kgc, b['cv'], b['tx'], kg_filler=filler)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        z = enc(b['drugs'], b['masks'], b['strs'], kgc, b['cv'], b['tx'], kg_filler=filler)
    torch.cuda.synchronize()
    print('encode ms', (time.perf_counter() - t) / 5 * 1e3)
