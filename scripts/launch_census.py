"""Which lines of madrigal_amd queue the launches of a steady-state finetune step: torch.profiler (with_stack) over one step after warm-up,
device kernels and memcpy / memset grouped by the innermost madrigal_amd frame that caused them (through the correlation id of the
launching runtime call).    python scripts/launch_census.py [--precision bf16] [--fresh-triples]"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madrigal_amd import configs, data, models as M  # noqa: E402
from madrigal_amd.optim import create_optimizer  # noqa: E402
from madrigal_amd.train import FinetuneStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--fresh-triples", action="store_true")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
N, L = 4096, 896
M.set_precision(a.precision)
batch, bkg = data.make_batch(N, seed=0, kg_nodes=130_000, kg_edges=8_000_000)
torch.manual_seed(0)
model = configs.build_model("twosides321", bkg["data"], n_outcomes=L).cuda()
b = data.batch_to(batch, "cuda")
kgc = {"data": bkg["data"].to("cuda"), "drug_index_map": bkg["drug_index_map"].cuda()}
trip = [tuple(t.cuda() for t in data.make_labelled_triples(N, L, 1_000_000, s)) for s in ((0, 1, 2, 3) if a.fresh_triples else (0,))]
filler = torch.randn(N, 128, device="cuda")
hp = dict(optimizer="adamw", structure_encoder_lr=1e-5, kg_encoder_lr=1e-5, perturb_encoders_lr=1e-5, fusion_lr=1e-6, decoder_lr=1e-4,
          wd=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
fs = FinetuneStep(model, create_optimizer(model, hp))
for i in range(3):
    lab, hd, tl, y = trip[i % len(trip)]
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
lab, hd, tl, y = trip[3 % len(trip)]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
    torch.cuda.synchronize()
evs = prof.events()
by_corr = {}
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and getattr(e, "stack", None):
        pass
# kineto: device events carry no stack; walk the CPU op events that enclose each launch (time containment on the same thread)
cpu_ops = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
dev = [e for e in evs if e.device_type != torch.autograd.DeviceType.CPU]
print(f"device events in the step: {len(dev)}; kernel time {sum(e.device_time_total if hasattr(e, 'device_time_total') else e.cuda_time_total for e in dev) / 1e3:.1f} ms")
kinds = collections.Counter()
for e in dev:
    n = e.name
    k = "torch " + n.split("<")[0].split("(")[0][-40:] if ("at::native" in n or "rocprim" in n or "Cijk" in n or "Memcpy" in n or "Memset" in n or "rocclr" in n) else "own"
    kinds[k] += 1
for k, c in kinds.most_common(30):
    print(f"{c:6d}  {k}")
# attribution by python frame: a TorchDispatchMode over one more step (backward on this thread) records, for every aten op that is
# not a pure view, the innermost madrigal_amd frame
from torch.utils._python_dispatch import TorchDispatchMode
VIEWS = ("view", "reshape", "_unsafe_view", "expand", "slice", "select", "t", "transpose", "permute", "unsqueeze", "squeeze", "detach", "alias", "as_strided",
         "narrow", "unbind", "split", "split_with_sizes", "unfold", "_reshape_alias", "empty", "empty_like", "empty_strided", "new_empty", "lift_fresh", "sym_size",
         "is_same_size", "_local_scalar_dense", "resize_", "set_", "stride", "size", "numel", "dim", "is_pinned", "record_stream")
src = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0] if hasattr(func, "__name__") else str(func)
        if name not in VIEWS:
            f = sys._getframe(1)
            where = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if "madrigal_amd" in fn:
                    where = f"{os.path.basename(fn)}:{f.f_lineno} {f.f_code.co_name}"
                    break
                f = f.f_back
            src[(where, name)] += 1
        return func(*args, **(kwargs or {}))


lab, hd, tl, y = trip[0]
with torch.autograd.set_multithreading_enabled(False), Census():
    fs.step(b, b, b["masks"], b["masks"], kgc, lab, hd, tl, y, kg_filler=filler)
torch.cuda.synchronize()
print(f"\naten ops (non-view) in one step: {sum(src.values())}; by innermost madrigal_amd frame:")
for (frame, name), c in src.most_common(a.top):
    print(f"{c:5d}  {name:24s} {frame}")
