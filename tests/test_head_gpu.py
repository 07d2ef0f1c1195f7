"""HIP all-pairs bilinear head vs the CPU oracle and the reference golden vectors (through the C ABI)."""
import numpy as np
import pytest
import torch

from helpers import rel_err, set_switch, t

pytestmark = pytest.mark.gpu

# tolerance per arithmetic mode, norm-wise relative (max|d| / max|ref|).  BASELINE.json: fp32 scores
# within 1e-4 relative of the reference CPU path; bf16 is the reduced-precision config ("bf16").
# "f16" = BASELINE configs[4] ("fp16 bilinear head"): operands rounded to IEEE half once, 8x finer than bf16.
TOL = {"f32": 2e-5, "bf16x3": 1e-4, "bf16": 3e-2, "f16": 4e-3}


@pytest.fixture(scope="module")
def ops():
    from madrigal_amd import ops as _ops
    return _ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _oracle(zh, zt, w):
    from oracle import madrigal_oracle as O
    return O.bilinear_scores(zh, zt, w)


def test_symmetrize(ops):
    from oracle import madrigal_oracle as O
    w = _rand((7, 128, 128), 0)
    out = ops.symmetrize(w.cuda())
    assert torch.equal(out.cpu(), O.symmetric(w))          # pure data movement: bit exact
    w2 = w.cuda()
    ops.symmetrize(w2, out=w2)                              # in place
    assert torch.equal(w2.cpu(), O.symmetric(w))


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16", "f16"])
def test_golden_head(ops, golden, prec):
    g = golden("head")
    zh, zt, w = t(g["z_head"]).cuda(), t(g["z_tail"]).cuda(), t(g["w_original"]).cuda()
    ws = ops.symmetrize(w)
    s = ops.bilinear_allpairs(zh, zt, ws, precision=prec).cpu()
    assert s.shape == (5, 24, 17)
    assert rel_err(s, g["scores"]) < TOL[prec]
    s14 = ops.bilinear_allpairs(zh, zt, ws[1:4], precision=prec).cpu()       # label_range=(1,4)
    assert rel_err(s14, g["scores_1_4"]) < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("nh,nt,L", [(256, 256, 32), (301, 77, 3), (1, 1, 1), (31, 513, 2), (700, 64, 2), (257, 129, 5), (130, 132, 3), (33, 4, 2)])
def test_vs_oracle(ops, prec, nh, nt, L):
    zh, zt = _rand((nh, 128), 1), _rand((nt, 128), 2)
    w = _rand((L, 128, 128), 3, 1 / np.sqrt(128))
    ref = _oracle(zh, zt, w)
    got = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision=prec).cpu()
    assert got.shape == ref.shape
    # scale floor sqrt(D): a lone score can sit near zero while its bf16 rounding error does not
    assert float((got - ref).abs().max()) < TOL[prec] * max(float(ref.abs().max()), 128 ** 0.5)


@pytest.mark.parametrize("prec,np_dtype", [("f16", np.float16), ("bf16", None)])
def test_rounded_operand_modes_against_the_rounded_oracle(ops, prec, np_dtype):
    """The single-product modes round z_head, W, T = z_head W (fp32) and z_tail to the 16-bit type once and accumulate in
    fp32: restating exactly those roundings in the oracle (float64 accumulation) pins the mode to fp32-accumulation noise,
    far below the mode's own distance from the fp32 result."""
    from oracle import madrigal_oracle as O
    zh, zt = _rand((130, 128), 1), _rand((257, 128), 2)
    w = _rand((3, 128, 128), 3, 1 / np.sqrt(128))
    ref = O.bilinear_scores_rounded(zh, zt, w, prec)
    got = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision=prec).cpu()
    # a T entry within fp32 rounding distance of a 16-bit tie may round the other way: allow a few such entries
    err = (got.double() - ref).abs() / float(ref.abs().max())
    assert float(err.median()) < 2e-7 and float(err.max()) < 2e-4, (float(err.median()), float(err.max()))
    assert rel_err(got, _oracle(zh, zt, w)) < TOL[prec]


@pytest.mark.parametrize("nh,nt", [(1000, 3004), (1000, 3001)])
def test_unaligned_output_view_is_written_in_place(ops, nh, nt):
    """A view that starts 4 bytes into an allocation (not 16-byte aligned): the same bits as into a fresh tensor, nothing before it."""
    zh, zt = _rand((nh, 128), 40).cuda(), _rand((nt, 128), 41).cuda()
    w = ops.symmetrize(_rand((5, 128, 128), 42, 1 / np.sqrt(128)).cuda())
    buf = torch.full((5 * nh * nt + 1,), float("nan"), device="cuda")
    view = buf[1:].view(5, nh, nt)
    ops.bilinear_allpairs(zh, zt, w, precision="bf16x3", out=view)
    assert torch.equal(view, ops.bilinear_allpairs(zh, zt, w, precision="bf16x3")) and bool(torch.isnan(buf[0]))


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("N,L", [(1024, 3), (1000, 2), (516, 2), (2304, 1), (1001, 2), (771, 2), (1283, 1)])
def test_symmetric_sweep_for_one_drug_set(ops, monkeypatch, prec, N, L):
    """decoder(z, z, ...) (predict.py:428): the same matrix on both sides runs the symmetric sweep -- tiles on / right of the
    block diagonal computed, every off-diagonal tile stored twice.  Against the general kernel (MDG_BILINEAR_SYMMETRIC=0):
    the computed half is the same arithmetic in the same order => identical bits; the mirrored half is its exact transpose
    and within the mode's tolerance of what the general kernel computes there (the other association order).  Ragged N
    (not a multiple of 256 / 64 / 4: rows of the score matrix then start at any 4-byte alignment and the sweep's 16-byte stores
    are unaligned; the last N % 4 columns come from a strip launch of the general kernel), odd numbers of row blocks and the
    sigmoid epilogue included."""
    z = _rand((N, 128), 60).cuda()
    w = ops.symmetrize(_rand((L, 128, 128), 61, 1 / np.sqrt(128)).cuda())
    set_switch(monkeypatch, "MDG_BILINEAR_SYMMETRIC", "0")
    gen = ops.bilinear_allpairs(z, z, w, precision=prec)
    gen_sig = ops.bilinear_allpairs(z, z, w, precision=prec, epilogue=ops.EPI_STORE_SIGMOID)
    set_switch(monkeypatch, "MDG_BILINEAR_SYMMETRIC", "1")
    out = torch.full((L, N, N), float("nan"), device="cuda")
    ops.bilinear_allpairs(z, z, w, precision=prec, out=out)
    assert not bool(torch.isnan(out).any())
    blk = torch.arange(N, device="cuda") // 256
    upper = (blk[None, :] >= blk[:, None])                                   # on / right of the block diagonal: computed
    scale = float(gen.abs().max())
    n4 = N - N % 4
    strip = torch.zeros(N, N, dtype=torch.bool, device="cuda")
    strip[n4:, :] = True                                                       # mirror images of the strip columns [n4, N)
    swept = upper & ~strip.T                                                   # computed by the sweep itself (not the column strip)
    if prec == "f32":                       # same instruction, same order: identical bits
        assert torch.equal(out[:, swept], gen[:, swept])
    else:                                   # 16-bit operand modes sweep on v_mfma 16x16x32 (general kernel: 32x32x16): same products,
        assert float((out[:, swept] - gen[:, swept]).abs().max()) < 2e-6 * scale      # fp32 sums grouped 32 instead of 16 deep
    lower = ~upper
    outT = out.transpose(1, 2)
    assert torch.equal(out[:, lower & ~strip], outT[:, lower & ~strip])        # the mirrored half: exact transpose
    if n4 != N:                                # the strip (its own small kernel, fp32 sums in another order) against the general kernel
        noise = {"f32": 2e-6, "bf16x3": 1e-5, "bf16": 1e-2, "f16": 1.5e-3}[prec] * scale       # and against its mirror image from the sweep
        assert float((out[:, :, n4:] - gen[:, :, n4:]).abs().max()) < noise
        assert float((out[:, lower & strip] - outT[:, lower & strip]).abs().max()) < noise
    assert float((out - gen).abs().max()) < max(TOL[prec], 1e-6) * scale
    sg = ops.bilinear_allpairs(z, z, w, precision=prec, epilogue=ops.EPI_STORE_SIGMOID)
    assert float((sg[:, swept] - gen_sig[:, swept]).abs().max()) < 2e-6 and torch.equal(sg[:, lower & ~strip], sg.transpose(1, 2)[:, lower & ~strip])
    # the row-pitched layout (ops.empty_scores: rows on 128-byte lines, no strip launch, whole-line stores): the sweep's own bits everywhere
    pit = ops.empty_scores(L, N, N, "cuda")
    pit.fill_(float("nan"))
    assert pit.stride(1) % 32 == 0 and ops.bilinear_allpairs(z, z, w, precision=prec, out=pit) is pit
    assert not bool(torch.isnan(pit).any())
    assert torch.equal(pit[:, ~strip & ~strip.T], out[:, ~strip & ~strip.T]) and torch.equal(pit[:, lower], pit.transpose(1, 2)[:, lower])
    assert float((pit - gen).abs().max()) < max(TOL[prec], 1e-6) * scale
    assert torch.equal(sg, torch.sigmoid(out)) or float((sg - 1.0 / (1.0 + torch.exp(-out))).abs().max()) < 1e-6
    # a different tensor with the same values is not "the same matrix": the general kernel runs, bit for bit
    assert torch.equal(ops.bilinear_allpairs(z, z.clone(), w, precision=prec), gen)
    ref = _oracle(z.cpu(), z.cpu(), _rand((L, 128, 128), 61, 1 / np.sqrt(128)))
    assert rel_err(out.cpu(), ref) < TOL[prec]


def test_symmetric_sweep_beyond_2GiB_per_outcome(ops, monkeypatch):
    """N = 23 552: one outcome's [N,N] slab is 2.2 GB, so the mirrored stores' byte offsets pass 2^31 (they are unsigned 32-bit
    offsets into a buffer descriptor of N*N*4 bytes).  Mirrored half == exact transpose; sampled blocks == the general kernel."""
    N = 23_552
    free, _ = torch.cuda.mem_get_info()
    if free < 6 * 2 ** 30:
        pytest.skip("needs 6 GB of free HBM")
    z = _rand((N, 128), 70).cuda()
    w = ops.symmetrize(_rand((1, 128, 128), 71, 1 / np.sqrt(128)).cuda())
    out = torch.full((1, N, N), float("nan"), device="cuda")
    ops.bilinear_allpairs(z, z, w, precision="bf16", out=out)
    s = out[0]
    for i0, j0 in ((0, 0), (N - 512, N - 512), (256, N - 300), (N - 300, 256), (11_776, 12_032), (20_000, 777)):
        blk = s[i0:i0 + 300, j0:j0 + 300]
        assert not bool(torch.isnan(blk).any())
        ref = ops.bilinear_allpairs(z[i0:i0 + 300].contiguous(), z[j0:j0 + 300].contiguous(), w, precision="bf16")[0]      # general kernel
        # blocks right of the block diagonal are computed (same products as the general kernel: fp32 grouping only); blocks
        # left of it are mirrored, i.e. the OTHER association order, which in a 16-bit mode differs at that mode's rounding of T
        computed = (i0 + 299) // 256 <= j0 // 256
        assert float((blk - ref).abs().max()) < (2e-6 if computed else TOL["bf16"]) * float(ref.abs().max()), (i0, j0)
        ib = (torch.arange(i0, i0 + 300, device="cuda") // 256)[:, None]
        jb = (torch.arange(j0, j0 + 300, device="cuda") // 256)[None, :]
        off = ib != jb                                       # outside the diagonal blocks: S[i,j] and S[j,i] are one stored value
        assert torch.equal(s[j0:j0 + 300, i0:i0 + 300].T[off], blk[off]), (i0, j0)
    assert bool(torch.isfinite(s.sum(dim=1)).all())            # every row written


def test_asymmetric_operands_catch_transposes(ops):
    """A = I style check with asymmetric operands: z_head one-hot rows pick out rows of W z_tail^T."""
    L, n = 2, 40
    zh = torch.zeros(n, 128)
    zh[torch.arange(n), torch.arange(n) * 3 % 128] = 1.0
    zt = _rand((50, 128), 5)
    w = _rand((L, 128, 128), 6)
    from oracle import madrigal_oracle as O
    ref = _oracle(zh, zt, w)
    got = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision="f32").cpu()
    assert rel_err(got, ref) < 1e-6
    assert rel_err(got, ref.transpose(1, 2)[:, :n, :50] if False else ref) < 1e-6


def test_sigmoid_epilogue(ops):
    zh, zt, w = _rand((100, 128), 7), _rand((90, 128), 8), _rand((3, 128, 128), 9, 0.05)
    ref = torch.sigmoid(_oracle(zh, zt, w))
    got = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision="f32",
                                epilogue=ops.EPI_STORE_SIGMOID).cpu()
    assert float((got - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("prec", ["f32", "bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("nh,nt", [(300, 333), (1, 1), (513, 64), (1100, 130)])
def test_rowstats_epilogue(ops, prec, nh, nt):
    """(bf16 / f16 run the two-row-blocks-per-wave variant: 512 head rows per workgroup, ragged row and column tails.)"""
    zh, zt, w = _rand((nh, 128), 10), _rand((nt, 128), 11), _rand((4, 128, 128), 12, 0.1)
    ref = _oracle(zh, zt, w)
    st = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision=prec,
                               epilogue=ops.EPI_ROWSTATS).cpu()
    assert st.shape == (4, nh, 2)
    scale = float(ref.abs().max())
    assert float((st[..., 1] - ref.max(dim=2).values).abs().max()) < TOL[prec] * scale
    assert float((st[..., 0] - ref.sum(dim=2)).abs().max()) < TOL[prec] * scale * max(nt, 2) ** 0.5 * 4
    dense = ops.bilinear_allpairs(zh.cuda(), zt.cuda(), ops.symmetrize(w.cuda()), precision=prec).cpu()
    if prec in ("f32", "bf16x3"):
        assert torch.equal(st[..., 1], dense.max(dim=2).values)      # same products, same order, same maxima, bit for bit
    else:       # 16-bit modes: the statistics run on the 16x16x32 instruction (32 products per fp32 add chain step, not 16)
        assert float((st[..., 1] - dense.max(dim=2).values).abs().max()) < 2e-6 * scale


def test_empty_and_errors(ops):
    z = _rand((4, 128), 0).cuda()
    w = _rand((2, 128, 128), 1).cuda()
    assert ops.bilinear_allpairs(z[:0], z, w).shape == (2, 0, 4)
    assert ops.bilinear_allpairs(z, z[:0], w).shape == (2, 4, 0)
    assert ops.bilinear_allpairs(z, z, w[:0]).shape == (0, 4, 4)
    with pytest.raises(ValueError):
        ops.bilinear_allpairs(z[:, :64], z, w)
    with pytest.raises(ValueError):
        ops.bilinear_allpairs(z.cpu(), z, w)
    with pytest.raises(ValueError):
        ops.bilinear_allpairs(z.double(), z, w)
    with pytest.raises(ValueError):
        ops.bilinear_allpairs(z, z, w, precision="fp8")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_full_size_properties(ops, prec):
    """BASELINE cfg2/4 size (N=4096, L=896 would be 60 GB: use L=64 here, same N): size-independent
    properties -- symmetry in (i,j) for z_head == z_tail, linearity in W, rowstats == reduction of the
    stored tensor -- plus random blocks against the oracle."""
    N, L = 4096, 64
    z = _rand((N, 128), 20)
    w = _rand((L, 128, 128), 21, 1 / np.sqrt(128))
    zc, wc = z.cuda(), ops.symmetrize(w.cuda())
    s = ops.bilinear_allpairs(zc, zc, wc, precision=prec)
    scale = float(s.abs().max())
    assert float((s - s.transpose(1, 2)).abs().max()) < TOL[prec] * scale
    s2 = ops.bilinear_allpairs(zc, zc, 2.0 * wc, precision=prec)
    assert float((s2 - 2.0 * s).abs().max()) < TOL[prec] * scale * 2
    st = ops.bilinear_allpairs(zc, zc, wc, precision=prec, epilogue=ops.EPI_ROWSTATS)
    assert float((st[..., 1] - s.max(dim=2).values).abs().max()) < TOL[prec] * scale
    assert float((st[..., 0] - s.sum(dim=2)).abs().max()) < TOL[prec] * scale * 64 * 4
    from oracle import madrigal_oracle as O
    rng = np.random.default_rng(0)
    for _ in range(6):
        l, i0, j0 = int(rng.integers(0, L)), int(rng.integers(0, N - 200)), int(rng.integers(0, N - 300))
        ref = O.bilinear_scores(z[i0:i0 + 200], z[j0:j0 + 300], w[l:l + 1])
        assert rel_err(s[l:l + 1, i0:i0 + 200, j0:j0 + 300].cpu(), ref) < TOL[prec]


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_cfg5_scale_row_statistics(ops, prec):
    """BASELINE configs[4] (100k drugs, 16-bit head; nothing can be materialised): N = 100 352 with a few outcomes through the
    row-statistics epilogue.  (i) the row-sum identity  sum_j S[l,i,j] = T[l,i,:] . sum_j z_j  with the mode's roundings
    restated (float64 sums); (ii) a block of head rows scored densely by the same kernel and by the CPU oracle: the
    statistics of the full run equal the reductions of that dense block (maxima bit for bit)."""
    from oracle import madrigal_oracle as O
    N, L = 100_352, 3
    z = _rand((N, 128), 50)
    w = _rand((L, 128, 128), 51, 1 / np.sqrt(128))
    zc, wc = z.cuda(), ops.symmetrize(w.cuda())
    st = ops.bilinear_allpairs(zc, zc, wc, precision=prec, epilogue=ops.EPI_ROWSTATS).cpu()
    assert st.shape == (L, N, 2) and bool(torch.isfinite(st).all())
    # (i) every row of every outcome
    zr = O.round_operand(z, prec).double()
    T = O.round_operand((zr @ O.round_operand(O.symmetric(w), prec).double()).float(), prec).double()     # [L,N,128]
    want = T @ zr.sum(dim=0)                                                                          # [L,N]
    scale = float(st[..., 1].abs().max()) * N ** 0.5                                                  # size of a random-sign row sum
    err = (st[..., 0].double() - want).abs() / scale
    # an entry of T within fp32 rounding distance of a 16-bit tie may round the other way on the device: that moves a row
    # sum by one 16-bit ulp of T times |sum_j z_j| -- rare, and bounded
    assert float(err.median()) < 2e-6 and float((err > 2e-4).double().mean()) < 1e-3 and float(err.max()) < 4e-3, \
        (float(err.median()), float((err > 2e-4).double().mean()), float(err.max()))
    # (ii) 96 head rows (ragged against the 512-row workgroups) x all 100 352 tails, densely
    rows = torch.arange(70_001, 70_097)
    dense = ops.bilinear_allpairs(zc[rows.cuda()], zc, wc, precision=prec).cpu()
    ref = O.bilinear_scores(z[rows], z, w)
    assert rel_err(dense, ref) < TOL[prec]
    # (statistics: v_mfma 16x16x32, dense: 32x32x16 -- same products, fp32 sums grouped differently)
    assert float((st[:, rows, 1] - dense.max(dim=2).values).abs().max()) < 2e-6 * float(dense.abs().max())
    assert float((st[:, rows, 0] - dense.sum(dim=2)).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
@pytest.mark.parametrize("nh,nt,L", [(4096, 4096, 24), (1000, 3001, 7)])
def test_repeated_launches_are_bit_identical(ops, prec, nh, nt, L):
    """Race detector: the kernel is deterministic (fixed summation order, no atomics), so every launch into a
    NaN-prefilled buffer must reproduce the first one bit for bit -- an LDS tile consumed before it landed or
    an unwritten element would show up here."""
    zh, zt = _rand((nh, 128), 40).cuda(), _rand((nt, 128), 41).cuda()
    w = ops.symmetrize(_rand((L, 128, 128), 42, 1 / np.sqrt(128)).cuda())
    first = None
    for it in range(6):
        out = torch.full((L, nh, nt), float("nan"), device="cuda")
        ops.bilinear_allpairs(zh, zt, w, precision=prec, out=out)
        assert not bool(torch.isnan(out).any())
        if first is None:
            first = out
        else:
            assert torch.equal(out, first), f"launch {it} differs from launch 0"


@pytest.mark.parametrize("N,L", [(4003, 3), (11607, 1)])
def test_real_drug_counts_on_the_symmetric_sweep(ops, N, L):
    """SURVEY 8(d)'s ragged input (4 003 drugs) and the drug count of the reference's own scoring run (11 607,
    generate_embeddings.ipynb): contiguous [L,N,N] (unaligned rows + column strip) and row-pitched output against the oracle on
    sampled head rows, and against each other."""
    z = _rand((N, 128), 70).cuda()
    w0 = _rand((L, 128, 128), 71, 1 / np.sqrt(128))
    w = ops.symmetrize(w0.cuda())
    out = ops.bilinear_allpairs(z, z, w, precision="bf16x3")
    pit = ops.bilinear_allpairs(z, z, w, precision="bf16x3", out=ops.empty_scores(L, N, N, "cuda"))
    assert out.is_contiguous() and not pit.is_contiguous()
    rows = torch.tensor([0, 1, 255, 256, 257, N // 2, N - 5, N - 4, N - 3, N - 2, N - 1])
    ref = _oracle(z.cpu()[rows], z.cpu(), w0)
    assert rel_err(out[:, rows.cuda()].cpu(), ref) < TOL["bf16x3"] and rel_err(pit[:, rows.cuda()].cpu(), ref) < TOL["bf16x3"]
    assert float((out - pit).abs().max()) < 1e-5 * float(out.abs().max())
    assert float((pit[0] - pit[0].T).abs().max()) < 1e-5 * float(out.abs().max())
