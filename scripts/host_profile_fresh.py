"""Host profile of the finetune step with a FRESH head batch, tail batch and modality masks every step (what bench.py's finetune leg times):
where the 66 ms beyond the cached-plan step go.  python scripts/host_profile_fresh.py"""
import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data as D, models as M
from madrigal_amd.optim import create_optimizer
from madrigal_amd.train import FinetuneStep
N, L = 4096, 896
batch, bkg = D.make_batch(N, 0, kg_nodes=130000, kg_edges=8000000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], L).cuda()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
sets = [tuple(t.cuda() for t in D.make_labelled_triples(N, L, 1000000, s)) for s in range(3)]
filler = torch.randn(N, 128, device="cuda")
avail = batch["masks"]
sides = [[D.batch_to(D.make_batch(N, sd + v, kg=bkg["data"], masks=avail)[0], "cuda") for sd in (200, 300)] for v in range(2)]
g = torch.Generator(device="cuda").manual_seed(99)
avail_dev = avail.cuda()
def draw():
    drop = torch.rand(avail_dev.shape, generator=g, device="cuda") < 0.3
    drop[:, 0] = False
    return avail_dev | drop
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4, wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
def step(i):
    hb, tb = sides[i % 2]
    return fs.step(hb, tb, draw(), draw(), kgc, *sets[i % 3], kg_filler=filler)
with M.precision("bf16"):
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        step(i)
    t_issue = (time.perf_counter() - t0) / 6
    torch.cuda.synchronize()
    print(f"fresh sides: host issue {t_issue * 1e3:.1f} ms per step, with sync {(time.perf_counter() - t0) / 6 * 1e3:.1f} ms per step")
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
print(s.getvalue()[:14000])
