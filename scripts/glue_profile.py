"""Which call sites of the finetune step launch torch's own small kernels (fill / add / copy / cat / index)?  torch profiler with
Python stacks; prints, per aten op, the innermost madrigal_amd frames and their counts per step."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from madrigal_amd import configs, data as D, models as M
from madrigal_amd.optim import create_optimizer
from madrigal_amd.train import FinetuneStep
M.set_precision("bf16")
batch, bkg = D.make_batch(4096, 0, kg_nodes=130000, kg_edges=8000000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], 896).cuda()
b = D.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
lab, hd, tl, y = (t.cuda() for t in D.make_labelled_triples(4096, 896, 1000000, 0))
filler = torch.randn(4096, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4, wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
for _ in range(2):
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
torch.cuda.synchronize()
import sys as _sys
from torch.overrides import TorchFunctionMode, resolve_name


class Sites(TorchFunctionMode):
    """Counts torch.* calls made from the package's Python (not the ones the autograd engine makes itself) by call site."""

    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_function__(self, func, types, args=(), kwargs=None):
        f = _sys._getframe(1)
        site = None
        while f is not None:
            fn = f.f_code.co_filename
            if "madrigal_amd" in fn:
                site = f"{os.path.basename(fn)}:{f.f_lineno}"
                break
            f = f.f_back
        name = resolve_name(func) or getattr(func, "__name__", str(func))
        self.count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


LAUNCHING = ("zeros", "zero_", "fill_", "ones", "full", "add", "sub", "mul", "div", "cat", "stack", "index_select", "clone", "copy_", "contiguous",
             "index_add", "sum", "cumsum", "arange", "where", "to", "reshape", "__getitem__", "__setitem__", "repeat_interleave", "gather",
             "scatter", "empty", "neg", "exp", "sigmoid", "mean", "masked_fill", "expand", "float", "long", "bool", "any", "all", "sort", "argsort")
with Sites() as sites:
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
    torch.cuda.synchronize()
by_op = collections.Counter()
for (name, site), c in sites.count.items():
    by_op[name] += c
print("torch calls from package Python in one step:", sum(by_op.values()))
for name, c in by_op.most_common(40):
    top = sorted(((cc, st) for (n, st), cc in sites.count.items() if n == name), reverse=True)[:6]
    print(f"== {name}: {c}   " + "  ".join(f"{st}x{cc}" for cc, st in top))
_sys.exit(0)
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::cat", "aten::index_select", "aten::index_add_", "aten::index", "aten::index_put_", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous")
agg = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if "madrigal_amd" in f or "autograd" in f.lower()]
        key = " <- ".join(f.split("/")[-1][:70] for f in frames[:2]) or "(engine / no python frame)"
        agg[ev.name][key] += 1
for op in want:
    tot = sum(agg[op].values())
    if tot:
        print(f"== {op}: {tot}")
        for k, c in agg[op].most_common(8):
            print(f"   {c:5d}  {k}")
