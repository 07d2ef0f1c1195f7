"""Head kernel variants at the bench shape (4096 x 4096 x 896): time per launch (HIP events), bit-identity across variants."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--labels", type=int, default=896)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--precisions", default="bf16x3,bf16,f16,f32")
ap.add_argument("--variants", default="0,2")
ap.add_argument("--sym", default="0,1")
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
z = torch.randn(a.n, 128, generator=g).cuda()
w = ops.symmetrize((torch.randn(a.labels, 128, 128, generator=g) / 128 ** 0.5).cuda())
out = torch.empty(a.labels, a.n, a.n, device="cuda")
stamps = torch.zeros(2 * a.labels * ((a.n + 255) // 256), dtype=torch.int64, device="cuda")       # in-kernel clock stamps (diagnostics)
import ctypes
from madrigal_amd._lib import lib
lib().mdg_debug_bilinear_stamps(ctypes.c_void_p(stamps.data_ptr()), ctypes.c_int64(stamps.numel()))
res = {}
for prec in a.precisions.split(","):
    ref = None
    for var in [v + "s" + y for y in a.sym.split(",") for v in (a.variants.split(",") if y == "0" else ["-"])]:
        os.environ["MDG_BILINEAR_VARIANT"] = var[0] if var[0] != "-" else "0"
        os.environ["MDG_BILINEAR_SYMMETRIC"] = var[-1]
        lib().mdg_tuning_reload()
        for _ in range(2):
            ops.bilinear_allpairs(z, z, w, precision=prec, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
        ev[0].record()
        for i in range(a.reps):
            ops.bilinear_allpairs(z, z, w, precision=prec, out=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
        chk = out[::37, ::5, ::3].clone()
        same = None if ref is None else (bool(torch.equal(chk, ref)) if var[-1] == "0" else float((chk - ref).abs().max() / ref.abs().max()))
        ref = chk if ref is None else ref
        st = stamps.view(-1, 2).double()
        st = st[st[:, 1] > 0]
        clk = float((st[:, 0] / st[:, 1]).median()) * 100.0 if st.numel() else 0.0                 # MHz (s_memrealtime ticks at 100 MHz)
        res[f"{prec}/v{var}"] = {"clock_mhz": round(clk), "sweep_cycles_median": float(st[:, 0].median()) if st.numel() else 0.0, "ms_median": ts[len(ts) // 2], "ms_min": ts[0], "tb_s": a.labels * a.n * a.n * 4 / ts[len(ts) // 2] / 1e9,
                                 "same_as_v0": same}
        print(prec, var, res[f"{prec}/v{var}"], flush=True)
print(json.dumps(res))
