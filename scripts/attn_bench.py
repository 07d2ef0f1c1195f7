"""Time the fusion self-attention forward / backward kernels at the finetune shape (compact live-token layout: ~5.4 live tokens
per drug, 32-row tiles, 8 heads x 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrigal_amd import configs, data, models as M, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch, bkg = data.make_batch(n, seed=0, kg_nodes=2000, kg_edges=20000)
model = configs.build_model("twosides321", bkg["data"], n_outcomes=8).cuda().eval()
enc = model.encoder
b = data.batch_to(batch, "cuda")
plan = enc._mask_plan(b["masks"], torch.device("cuda"), True)["live"]
tf = enc.transformer
H, dh, S = tf.num_heads, tf.head_dim, plan["S"]
R = plan["R"]
qkv = torch.randn(R, 3 * H * dh, device="cuda")
dout = torch.randn(R, H * dh, device="cuda")
print(f"rows {R}, tiles {plan['n_tiles']}, heads {H} x {dh}")
def run(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e6
fw = run(lambda: ops.fusion_attention(qkv, plan["n_tiles"], S, H, dh, row_start=plan["tile_start"], row_bits=plan["row_bits"]))
bw = run(lambda: ops.fusion_attention_bwd(qkv, dout, plan["n_tiles"], S, H, dh, row_start=plan["tile_start"], row_bits=plan["row_bits"]))
gb_f = (qkv.numel() + dout.numel()) * 4 / 1e9
gb_b = (2 * qkv.numel() + dout.numel()) * 4 / 1e9
print(f"forward {fw:.0f} us ({gb_f / fw * 1e6 / 1e3:.2f} TB/s of qkv+out), backward {bw:.0f} us ({gb_b / bw * 1e6 / 1e3:.2f} TB/s of qkv+dout+dqkv)")
