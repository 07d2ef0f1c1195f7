"""128-tile dense block with x staged straight from its fp32 rows (no pre-pass) against the pre-pass path: bit-identity on ragged
shapes in both 16-bit modes (incl. bias / activation / residual and a strided x), and the time of both.  python scripts/rawx_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
from madrigal_amd._lib import lib


def run(flag, fn):
    os.environ["MDG_LINEAR_RAWX"] = flag; lib().mdg_tuning_reload()
    out = fn()
    os.environ.pop("MDG_LINEAR_RAWX"); lib().mdg_tuning_reload()
    return out


torch.manual_seed(0)
for prec in ("bf16x3", "bf16"):
    for (M, N, K) in [(1000, 128, 68), (130000, 128, 128), (4097, 384, 132), (257, 100, 64), (50000, 256, 256), (3000, 128, 1000), (127, 2048, 128)]:
        xbig = torch.randn(M, K + 8, device="cuda")
        x = xbig[:, 4:4 + K] if K % 4 == 0 and False else torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
        f = lambda: ops.linear(x, w, b, act="relu", residual=r, precision=prec)
        y1, y0 = run("1", f), run("0", f)
        ref = torch.relu(x.double() @ w.double().T + b.double()) + r.double()
        err = float((y1.double() - ref).abs().max() / ref.abs().max())
        print(f"{prec} M={M} N={N} K={K}: raw-x == pre-pass bits: {bool(torch.equal(y0, y1))}; vs fp64 {err:.2e}", flush=True)
        assert torch.equal(y0, y1) and err < (3e-2 if prec == "bf16" else 2e-5)
    # strided rows (a column slice of a wider tensor)
    xw = torch.randn(5000, 256, device="cuda"); xs = xw[:, 64:192]
    w = torch.randn(128, 128, device="cuda") / 11
    assert torch.equal(run("1", lambda: ops.linear(xs, w, precision=prec)), run("0", lambda: ops.linear(xs, w, precision=prec)))
print("bit-identical")
for prec in ("bf16x3", "bf16"):
    for (M, N, K) in [(130000, 128, 128), (106000, 128, 68), (130000, 384, 128), (20480, 2048, 128), (50000, 256, 256), (65536, 128, 512)]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
        res = []
        for flag in ("0", "1"):
            os.environ["MDG_LINEAR_RAWX"] = flag; lib().mdg_tuning_reload()
            for _ in range(3):
                ops.linear(x, w, precision=prec)
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.linear(x, w, precision=prec); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append(sorted(ts)[4] * 1e3)
        os.environ.pop("MDG_LINEAR_RAWX"); lib().mdg_tuning_reload()
        print(f"{prec} M={M:6d} N={N:5d} K={K:4d}: pre-pass + kernel {res[0]:7.1f} us   raw-x kernel {res[1]:7.1f} us", flush=True)
