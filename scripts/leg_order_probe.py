"""Does a training leg run slower when another one ran before it in the same process?  finetune steps, then contrastive steps (or the
reverse), on one model; prints ms per step of each.   python scripts/leg_order_probe.py [ft,pt | pt,ft | pt | ft] [--clear]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from madrigal_amd import configs, data as D, masks as MK, models as M, ops, autograd as ag
from madrigal_amd.optim import AdamW, create_optimizer
from madrigal_amd.simclr import SimCLR_NovelDDI
from madrigal_amd.train import FinetuneStep, PretrainStep

order = (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "ft,pt").split(",")
batch, bkg = D.make_batch(4096, 0, kg_nodes=130_000, kg_edges=8_000_000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], 896).cuda()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}


def finetune():
    lab, hd, tl, y = (t.cuda() for t in D.make_labelled_triples(4096, 896, 1_000_000, 0))
    filler = torch.randn(4096, 128, device="cuda")
    hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4, wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
    with M.precision("bf16"):
        fs = FinetuneStep(model, create_optimizer(model, hp))
        for i in range(25):
            if i == 5:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
        torch.cuda.synchronize()
    print(f"finetune: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step", flush=True)


def pretrain():
    B = 2048
    avail = batch["masks"][:B].clone().cpu()
    avail[:, 2] = torch.where(avail[:, 1:].all(dim=1), torch.zeros(B, dtype=torch.bool), avail[:, 2])
    pb, _ = D.make_batch(B, 0, kg=bkg["data"], masks=avail)
    np.random.seed(0)
    sim = SimCLR_NovelDDI(model.encoder, dim=128, mlp_dim=512, T=0.1, raw_encoder_output=True).cuda().train()
    pbd = D.batch_to(pb, "cuda")
    draw = MK.StrCenterUniSampler(MK.get_pretrain_masks(list(range(B)), avail.numpy().astype(np.int64), "str_center_uni", False, 0.2))
    step = PretrainStep(sim, AdamW(sim.parameters(), lr=1e-5, weight_decay=1e-2))
    data = (pbd["strs"], kgc, pbd["cv"], pbd["tx"])
    with M.precision("bf16x3"):
        for i in range(34):
            if i == 4:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            m1, m2 = draw(range(B))
            step.step(pb["drugs"], m1, m2, None, data)
        torch.cuda.synchronize()
    print(f"pretrain: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms per step", flush=True)


for leg in order:
    (finetune if leg == "ft" else pretrain)()
    if "--clear" in sys.argv:
        gc.collect(); torch.cuda.empty_cache(); ops._ws_cache.clear(); ag._gather_plan.clear()
