"""Dense-block timing with the phases separated (HIP events on the launch stream): operand pack of x alone, the GEMM kernel
alone on a packed x (ops.linear_packed), and the whole ops.linear call; plus a check of the 256-tile kernel against fp64 on
sampled rows / columns and bit-identity with the 128-tile kernel on ragged shapes.   python scripts/gemm_bench2.py [bf16|bf16x3] [check]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import ops
from madrigal_amd._lib import lib

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
P = ops._prec(prec)


def pack(x):
    M, K = x.shape
    nb = int(lib().mdg_pack_operand_bytes(ctypes.c_int64(M), ctypes.c_int64(K), ctypes.c_int(P)))
    img = torch.empty(nb, dtype=torch.uint8, device=x.device)
    ops.check(lib().mdg_pack_operand(ops._ptr(x), ctypes.c_int64(x.stride(0)), ctypes.c_int64(M), ctypes.c_int64(K), ctypes.c_int(P), ops._ptr(img),
                                     ctypes.c_size_t(nb), ops._stream(x)), "pack")
    return img


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


if "check" in sys.argv:
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K) in [(1024, 256 * 48, 64), (1500, 256 * 40 + 36, 100), (4096 + 77, 3072 + 4, 2048), (22464, 2048, 2048), (2048, 6144, 128), (1024 * 50, 256, 192)]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
        def run(tile, sk):
            os.environ["MDG_LINEAR_TILE"] = tile; os.environ["MDG_LINEAR_STREAMK"] = sk; lib().mdg_tuning_reload()
            return ops.linear(x, w, b, act="gelu", residual=r, precision=prec, cache_weight=False)
        y0 = run("128", "0")
        y1 = run("256", "0")                 # one tile per workgroup: the 128-tile kernel's sums, bit for bit
        y2 = run("256", "1")                 # stream-K hybrid (where the shape takes it): split tiles group their sums differently
        y3 = run("256", "1")
        for k_ in ("MDG_LINEAR_TILE", "MDG_LINEAR_STREAMK"):
            os.environ.pop(k_)
        lib().mdg_tuning_reload()
        rows = torch.randint(0, M, (64,), device="cuda"); rows[0] = M - 1; rows[1] = 0
        ref = torch.nn.functional.gelu(x[rows].double() @ w.double().T + b.double()) + r[rows].double()
        err = float((y2[rows].double() - ref).abs().max() / ref.abs().max())
        same = bool(torch.equal(y0, y1))
        sk_diff = float((y2 - y1).abs().max() / y1.abs().max())
        repro = bool(torch.equal(y2, y3))
        changed = float((y2 != y1).float().mean())
        worst = max(worst, err)
        print(f"{prec} M={M} N={N} K={K}: vs fp64 rel err {err:.2e}; 256-tile == 128-tile bits: {same}; stream-K vs plain: max rel diff {sk_diff:.1e} "
              f"({changed:.1%} of entries differ), reproducible: {repro}", flush=True)
        assert same and repro and sk_diff < 2e-6 and err < (3e-2 if prec == "bf16" else 2e-5)
    print("check ok")
    sys.exit(0)

shapes = [(22464, 2048, 2048), (20480, 6144, 2048), (20480, 2048, 2048), (20480, 1024, 2048), (20480, 2048, 1024), (6144, 2048, 20480), (65536, 512, 1024), (8192, 8192, 8192)]
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    img = pack(x)
    t_pack = timed(lambda: pack(x))
    t_kern = timed(lambda: ops.linear_packed(img, M, w, precision=prec))
    t_all = timed(lambda: ops.linear(x, w, precision=prec))
    fl = 2.0 * M * N * K * (3 if prec == "bf16x3" else 1)
    print(f"{prec} M={M:6d} N={N:5d} K={K:5d}  pack {t_pack[0]*1e3:7.1f} us  kernel {t_kern[0]*1e3:7.1f} us (min {t_kern[1]*1e3:7.1f}) = {fl/t_kern[0]/1e9:7.0f} TF of bf16 MFMA"
          f"   linear() {t_all[0]*1e3:7.1f} us = {fl/t_all[0]/1e9:7.0f} TF", flush=True)
