"""GPU: kernel-level parity of the C-ABI entry points (called through the ctypes binding) against fp64/fp32 torch-CPU
arithmetic and the oracle, at small odd sizes (edge cases: ragged tiles, empty inputs, gathers, ties) and at
BASELINE-sized shapes through size-independent properties."""
import math

import numpy as np
import pytest
import torch

from golden_util import close

pytestmark = pytest.mark.gpu
DEV = 'cuda'


import importlib as _importlib


class _LibProxy:
    """tests set _lib.CALL_LOG on the package's _lib module (resolved lazily: the package name is not an identifier)"""
    def _m(self):
        return _importlib.import_module('sibrar---single-branch-recommender_amd._lib')
    def __getattr__(self, k):
        return getattr(self._m(), k)
    def __setattr__(self, k, v):
        setattr(self._m(), k, v)


_lib = _LibProxy()


def S():
    import sibrar_amd
    return sibrar_amd


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (5, 7, 3), (33, 65, 31), (128, 64, 32), (257, 130, 100), (300, 64, 768), (1000, 200, 65)])
@pytest.mark.parametrize('act', [0, 1, 2])
def test_gemm_nt(M, N, K, act):
    ops = S().ops
    x, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2), _rand(N, seed=3)
    y = ops.linear_nt(x.to(DEV), w.to(DEV), b.to(DEV), act)
    pre = x.double() @ w.double().t() + b.double()
    ref = {0: pre, 1: torch.relu(pre), 2: torch.tanh(pre)}[act]
    # an fp32 dot product of K terms carries an absolute error ~ eps * sum|a_k b_k|, whatever the summation order; the
    # tolerance is therefore relative to the pre-activation scale (the activation has slope <= 1)
    close(y.cpu(), ref, rtol=1e-4, atol=1e-5, what='nt', norm_rtol=2e-6, scale=float(pre.abs().max()))


def test_gemm_nt_gather_scatter_unaligned():
    ops = S().ops
    table = _rand(50, 18, seed=4)                      # K = 18: not a multiple of 4 -> scalar load path
    w, b = _rand(8, 18, seed=5), _rand(8, seed=6)
    rows = torch.tensor([3, 3, 49, 0, 17, 21, 8], dtype=torch.int32)
    slots = torch.tensor([6, 0, 2, 9, 4, 1, 7], dtype=torch.int32)
    out = torch.full((10, 8), 7.0, device=DEV)
    ops.linear_nt(table.to(DEV), w.to(DEV), b.to(DEV), 1, a_idx=rows.to(DEV), out=out, c_idx=slots.to(DEV), n_rows=7)
    ref = torch.full((10, 8), 7.0, dtype=torch.float64)
    ref[slots.long()] = torch.relu(table[rows.long()].double() @ w.double().t() + b.double())
    close(out.cpu(), ref, rtol=1e-4, atol=1e-5, what='gather/scatter')


@pytest.mark.parametrize('M,N,K,act', [(1408, 128, 768, 1), (64, 128, 1024, 0), (300, 200, 512, 2), (7, 64, 2048, 1)])
def test_gemm_nt_split_k_with_gather_scatter(M, N, K, act):
    """Few output tiles and a long K (the modality projectors at the reference's default batch): sbr_gemm_nt_splitk_f32 — K
    split over workgroups, fixed-order reduction with bias, activation and output row scatter — against fp64; two runs are
    bit-identical."""
    ops = S().ops
    from importlib import import_module
    _lib = import_module(ops.__name__.rsplit('.', 1)[0] + '._lib')
    assert _lib.lib().sbr_gemm_nt_splitk_workspace(M, N, K) > 0
    table = _rand(M + 50, K, seed=31)
    w, b = _rand(N, K, seed=32) / math.sqrt(K), _rand(N, seed=33)
    g = torch.Generator().manual_seed(34)
    rows = torch.randint(0, M + 50, (M,), generator=g, dtype=torch.int32)
    slots = torch.randperm(M + 9, generator=g)[:M].to(torch.int32)
    outs = []
    for _ in range(2):
        out = torch.full((M + 9, N), 3.0, device=DEV)
        ops.linear_nt(table.to(DEV), w.to(DEV), b.to(DEV), act, a_idx=rows.to(DEV), out=out, c_idx=slots.to(DEV), n_rows=M)
        outs.append(out.cpu())
    pre = table[rows.long()].double() @ w.double().t() + b.double()
    ref = torch.full((M + 9, N), 3.0, dtype=torch.float64)
    ref[slots.long()] = {0: pre, 1: torch.relu(pre), 2: torch.tanh(pre)}[act]
    close(outs[0], ref, rtol=1e-4, atol=1e-5, what='split-K NT', norm_rtol=2e-6, scale=float(pre.abs().max()))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('M,N,K', [(3, 5, 7), (70, 33, 129), (513, 64, 64), (100, 768, 64)])
def test_gemm_nn_and_tn(M, N, K):
    ops = S().ops
    dz, w = _rand(M, K, seed=7), _rand(K, N, seed=8)
    close(ops.matmul_nn(dz.to(DEV), w.to(DEV)).cpu(), dz.double() @ w.double(), rtol=1e-4, atol=1e-5, what='nn', norm_rtol=1e-6)
    x = _rand(M, N, seed=9)
    got = ops.matmul_tn(dz.to(DEV), x.to(DEV))          # [K, N] = dz^T x over M rows
    close(got.cpu(), dz.double().t() @ x.double(), rtol=1e-4, atol=1e-5, what='tn', norm_rtol=1e-5)


def test_gemm_tn_large_reduction_with_gather():
    ops = S().ops
    R, C, F = 20000, 64, 96
    dz, table = _rand(R, C, seed=10), _rand(500, F, seed=11)
    rows = torch.randint(0, 500, (R,), generator=torch.Generator().manual_seed(12), dtype=torch.int32)
    got = ops.matmul_tn(dz.to(DEV), table.to(DEV), b_idx=rows.to(DEV), n_rows=R)
    ref = dz.double().t() @ table[rows.long()].double()
    close(got.cpu(), ref, rtol=1e-4, atol=1e-4, what='tn split-k', norm_rtol=1e-5)


def test_gemm_empty():
    ops = S().ops
    y = ops.linear_nt(torch.empty(0, 8, device=DEV), torch.randn(4, 8, device=DEV), None, 0)
    assert y.shape == (0, 4)
    g = ops.matmul_tn(torch.empty(0, 4, device=DEV), torch.empty(0, 8, device=DEV))
    assert torch.count_nonzero(g) == 0


@pytest.mark.parametrize('n,D', [(2, 3), (11, 7), (1000, 64), (4097, 130)])
@pytest.mark.parametrize('act', [0, 1, 3, 4])
def test_batchnorm_train_and_eval(n, D, act):
    ops = S().ops
    x = _rand(n, D, seed=20) * 2 + 0.5
    w, b = torch.rand(D) + 0.5, _rand(D, seed=21)
    r = _rand(n, D, seed=22)
    names = {0: None, 1: 'relu', 3: 'sigmoid', 4: 'selu'}
    from oracle import model_ref
    sd = {'weight': w.clone().requires_grad_(True), 'bias': b.clone().requires_grad_(True),
          'running_mean': torch.zeros(D), 'running_var': torch.ones(D), 'num_batches_tracked': torch.zeros((), dtype=torch.long)}
    xr = x.clone().requires_grad_(True)
    yr = model_ref._act(names[act], model_ref.batch_norm(xr, sd, '', True))
    (yr * r).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rm, rv, nb = torch.zeros(D, device=DEV), torch.ones(D, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
    y = ops.BatchNormActFn.apply(xd, wd, bd, rm, rv, nb, act)
    close(y.detach().cpu(), yr.detach(), rtol=1e-4, atol=1e-5, what='y')
    (y * r.to(DEV)).sum().backward()
    sc = float(xr.grad.abs().max())
    close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5, what='dx', norm_rtol=1e-4, scale=sc)
    close(wd.grad.cpu(), sd['weight'].grad, rtol=1e-4, atol=1e-4, what='dw', norm_rtol=1e-4)
    close(bd.grad.cpu(), sd['bias'].grad, rtol=1e-4, atol=1e-4, what='db', norm_rtol=1e-4)
    close(rm.cpu(), sd['running_mean'], rtol=1e-4, atol=1e-6, what='running_mean')
    close(rv.cpu(), sd['running_var'], rtol=1e-4, atol=1e-6, what='running_var')
    assert int(nb) == 1
    ye = ops.batch_norm_eval(x.to(DEV), w.to(DEV), b.to(DEV), rm, rv, act)
    yre = model_ref._act(names[act], model_ref.batch_norm(x, {k: v.detach() for k, v in sd.items()}, '', False))
    close(ye.cpu(), yre, rtol=1e-4, atol=1e-5, what='eval')


def test_l2norm_dropout_aggregate():
    ops = S().ops
    x = _rand(37, 24, seed=30)
    x[3] = 0                                           # zero row: clamped by eps
    xd = x.to(DEV).requires_grad_(True)
    y = ops.L2NormalizeFn.apply(xd)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.normalize(xr, p=2, dim=-1)
    close(y.detach().cpu(), yr.detach(), what='normalize')
    r = _rand(37, 24, seed=31)
    (y * r.to(DEV)).sum().backward()
    (yr * r).sum().backward()
    mask = torch.ones(37, dtype=torch.bool)
    mask[3] = False
    close(xd.grad.cpu()[mask], xr.grad[mask], rtol=1e-4, atol=1e-5, what='normalize grad', norm_rtol=1e-5)
    # dropout: keep-rate, scaling, backward uses the same mask
    big = torch.ones(1000, 64, device=DEV, requires_grad=True)
    d = ops.DropoutFn.apply(big, 0.2, 1234)
    kept = (d > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01 and torch.allclose(d[d > 0], torch.tensor(1.25, device=DEV))
    d.sum().backward()
    assert torch.equal(big.grad > 0, d > 0)
    assert not torch.equal(ops.DropoutFn.apply(big, 0.2, 1235) > 0, d > 0)
    # aggregate
    e = _rand(13, 2, 9, seed=32)
    for mode, fn in [(0, lambda t: t.mean(1)), (1, lambda t: t.max(1).values)]:
        ed = e.to(DEV).requires_grad_(True)
        er = e.clone().requires_grad_(True)
        out = ops.AggregateFn.apply(ed, mode)
        close(out.detach().cpu(), fn(er).detach(), what='aggregate')
        g = _rand(13, 9, seed=33)
        (out * g.to(DEV)).sum().backward()
        (fn(er) * g).sum().backward()
        close(ed.grad.cpu(), er.grad, what='aggregate grad')


@pytest.mark.parametrize('kind', ['bce', 'bpr', 'sampled_softmax'])
@pytest.mark.parametrize('agg', ['mean', 'sum'])
@pytest.mark.parametrize('B,N', [(1, 2), (6, 4), (300, 11), (8192, 11)])
def test_rec_losses_vs_oracle(kind, agg, B, N):
    import sibrar_amd as Sm
    from oracle import losses_ref
    logits = _rand(B, N, seed=40) * 3
    labels = torch.zeros(B, N, dtype=torch.float64)
    labels[:, 0] = 1
    for strat in (['uniform', 'uniform_recbole'] if kind == 'sampled_softmax' else ['uniform_recbole']):
        lr = logits.clone().requires_grad_(True)
        ref = losses_ref.RefRecLoss(kind, n_items=5000, aggregator=agg, train_neg_strategy=strat, neg_train=N - 1).compute_loss(lr, labels)
        ref.backward()
        ld = logits.to(DEV).requires_grad_(True)
        got = Sm.RecommenderSystemLossesEnum[kind].value(n_items=5000, aggregator=agg, train_neg_strategy=strat,
                                                         neg_train=N - 1).compute_loss(ld, labels.to(DEV))
        assert got.dtype == ref.dtype
        close(got.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6, what='loss')
        got.backward()
        close(ld.grad.cpu(), lr.grad, rtol=1e-4, atol=1e-7, what='dlogits', norm_rtol=1e-5)


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('B,N', [(1, 2), (300, 11), (8192, 11), (100000, 3)])
def test_rec_loss_one_launch_entry_point(kind, B, N):
    """sbr_rec_loss_fwd_bwd_ws (the fused step's loss: no zeroing launch, block partial sums through a self-resetting workspace,
    fixed summation order, packed (total, rec, reg) scalars) against sbr_rec_loss_fwd_bwd (itself pinned against the oracle above):
    same gradient bits, loss equal to rounding, the same loss bits on repeated calls, a dirty loss_out, and the workspace left zeroed."""
    from importlib import import_module
    _lib = import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    logits = (_rand(B, N, seed=44) * 3).to(DEV)
    labels = torch.zeros(B, N, dtype=torch.float64)
    labels[:, 0] = 1
    labels = labels.to(DEV)
    scale, shift = 1.0 / B, 0.37 if kind == 2 else 0.0
    l0, g0 = torch.full((1,), 7.0, device=DEV, dtype=torch.float64), torch.empty(B, N, device=DEV)
    _lib.call('sbr_rec_loss_fwd_bwd', kind, logits.data_ptr(), labels.data_ptr(), B, N, scale, shift, l0.data_ptr(), g0.data_ptr(),
              _lib.stream())
    need = int(_lib.lib().sbr_rec_loss_workspace(B))
    ws = torch.zeros(need // 8, device=DEV, dtype=torch.float64)
    seen = []
    for rep in range(3):
        l1, g1 = torch.full((1,), -3.0 * rep, device=DEV, dtype=torch.float64), torch.empty(B, N, device=DEV)
        out3 = torch.full((3,), 9.0, device=DEV, dtype=torch.float64)
        _lib.call('sbr_rec_loss_fwd_bwd_ws', kind, logits.data_ptr(), labels.data_ptr(), B, N, scale, shift, l1.data_ptr(),
                  g1.data_ptr(), out3.data_ptr() if rep != 1 else None, ws.data_ptr(), need, _lib.stream())
        assert torch.equal(g1, g0)
        assert abs(l1.item() - l0.item()) <= 1e-12 * abs(l0.item()) + 1e-15
        if rep != 1:
            assert out3.cpu().tolist() == [l1.item(), l1.item(), 0.0]
        assert ws.view(torch.int64)[0].item() == 0              # the arrival counter is reset
        seen.append(l1.item())
    assert seen[0] == seen[1] == seen[2]


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('B,N,D', [(1, 2, 64), (37, 11, 128), (8192, 11, 128), (3000, 16, 64), (5000, 5, 256), (70000, 3, 128)])
def test_fused_scorer_loss_statistics_kernel(kind, B, N, D):
    """sbr_bn_score_loss_fwd_bwd (scorer forward + recommendation loss + first backward pass of the fused tail in one launch)
    against the sequence it replaces — sbr_bn_score_fwd, sbr_rec_loss_fwd_bwd, sbr_bn_score_bwd_stats (each pinned on its own
    above / in the golden tests): logits, dlogits and dU bit for bit, the BatchNorm column sums and the loss to rounding, the packed
    loss scalars, a second call on the same (self-resetting) workspaces with the same bits."""
    from importlib import import_module
    _lib = import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    L = _lib.lib()
    assert L.sbr_bn_score_loss_supported(D, N) == 1
    z, u = (_rand(B * N, D, seed=61) * 2).to(DEV), _rand(B, D, seed=62).to(DEV)
    mean, rstd = (_rand(D, seed=63) * 0.1).to(DEV), (_rand(D, seed=64).abs() + 0.5).to(DEV)
    w, beta = (_rand(D, seed=65) * 0.5 + 1).to(DEV), (_rand(D, seed=66) * 0.1).to(DEV)
    labels = torch.zeros(B, N, dtype=torch.float64)
    labels[:, 0] = 1
    labels = labels.to(DEV)
    scale, shift, st = 1.0 / B, (0.37 if kind == 2 else 0.0), _lib.stream()
    KD = 2 * D
    # the three-kernel sequence
    ws0 = torch.zeros(17 * KD, device=DEV, dtype=torch.float64)
    lg0, dl0, du0 = torch.empty(B, N, device=DEV), torch.empty(B, N, device=DEV), torch.empty(B, D, device=DEV)
    l0 = torch.zeros(1, device=DEV, dtype=torch.float64)
    _lib.call('sbr_bn_score_fwd', z.data_ptr(), u.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), beta.data_ptr(),
              lg0.data_ptr(), B, N, D, st)
    _lib.call('sbr_rec_loss_fwd_bwd', kind, lg0.data_ptr(), labels.data_ptr(), B, N, scale, shift, l0.data_ptr(), dl0.data_ptr(), st)
    _lib.call('sbr_bn_score_bwd_stats', dl0.data_ptr(), u.data_ptr(), z.data_ptr(), du0.data_ptr(), B, N, D, w.data_ptr(),
              beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws0.data_ptr(), st)
    # the fused kernel, twice on the same workspaces
    ws1 = torch.zeros(17 * KD, device=DEV, dtype=torch.float64)
    lws = torch.zeros(int(L.sbr_bn_score_loss_workspace()) // 8, device=DEV, dtype=torch.float64)
    seen = []
    for rep in range(2):
        lg1, dl1, du1 = torch.full((B, N), 5.0, device=DEV), torch.empty(B, N, device=DEV), torch.empty(B, D, device=DEV)
        l1, out3 = torch.full((1,), 9.0, device=DEV, dtype=torch.float64), torch.full((3,), 9.0, device=DEV, dtype=torch.float64)
        _lib.call('sbr_bn_score_loss_fwd_bwd', z.data_ptr(), u.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(),
                  beta.data_ptr(), kind, labels.data_ptr(), scale, shift, lg1.data_ptr() if rep == 0 else None, dl1.data_ptr(),
                  du1.data_ptr(), l1.data_ptr(), out3.data_ptr(), B, N, D, ws1.data_ptr(), lws.data_ptr(), lws.numel() * 8, st)
        if rep == 0:
            assert torch.equal(lg1, lg0)
        assert torch.equal(dl1, dl0)
        assert torch.equal(du1, du0)
        assert abs(l1.item() - l0.item()) <= 1e-12 * abs(l0.item()) + 1e-15
        assert out3.cpu().tolist() == [l1.item(), l1.item(), 0.0]
        tot0, tot1 = ws0[:KD].cpu(), ws1[:KD].cpu()
        mag = tot0.abs().max().item() + 1e-30
        assert bool(((tot1 - tot0).abs() <= 1e-9 * mag + 1e-12 * B).all())
        assert bool((ws1[KD:] == 0).all()) and lws.view(torch.int64)[0].item() == 0      # replicas and counter left zeroed
        seen.append((l1.item(), tot1.clone()))
    assert seen[0][0] == seen[1][0]


def test_infonce_small_groups_read_in_place_from_the_slot_tensor():
    """The training step's call (engine: the two modality slices e[:, 0, :] / e[:, 1, :] of the [S, 2, D] embedding tensor read in
    place, ld = 2 D; gradients written into a [S, 2, D] tensor the same way) on the one-wave-per-group kernel: Onion18's group shape
    (N = 11, D = 128) with a group count that is no multiple of the four groups of a workgroup."""
    import sibrar_amd as Sm
    from oracle import model_ref
    ops = Sm.ops
    G, N, D = 1027, 11, 128
    e = _rand(G * N, 2, D, seed=52) * 0.5
    ar, br = e[:, 0, :].reshape(G, N, D).clone().requires_grad_(True), e[:, 1, :].reshape(G, N, D).clone().requires_grad_(True)
    ref = model_ref.info_nce(ar, br, 0.3)
    ref.backward()
    ed = e.to(DEV)
    loss = torch.zeros((), device=DEV, dtype=torch.float64)
    scale = 1.0 / (G * N)
    ops.infonce_fwd(ed.data_ptr(), ed.data_ptr() + 4 * D, 2 * D, G, N, D, 0.3, scale, loss, DEV)
    close(loss.cpu(), ref.detach().double(), rtol=1e-5, atol=1e-7, what='loss')
    de = torch.full_like(ed, float('nan'))
    gout = torch.ones((), device=DEV)
    ops.infonce_bwd(ed.data_ptr(), ed.data_ptr() + 4 * D, 2 * D, G, N, D, 0.3, scale, gout, de.data_ptr(), de.data_ptr() + 4 * D, 2 * D, DEV)
    close(de[:, 0, :].cpu().reshape(G, N, D), ar.grad, rtol=2e-4, atol=1e-7, what='da', norm_rtol=1e-4)
    close(de[:, 1, :].cpu().reshape(G, N, D), br.grad, rtol=2e-4, atol=1e-7, what='db', norm_rtol=1e-4)


@pytest.mark.parametrize('G,N,D', [(1, 2, 3), (7, 11, 16), (3, 101, 64), (1, 176, 8), (64, 40, 16), (1, 256, 64), (2, 300, 30),
                                   (1, 1000, 128), (4096, 11, 128), (5, 16, 256), (9, 1, 4), (130, 3, 64), (6, 17, 128)])
def test_infonce_vs_oracle(G, N, D):
    import sibrar_amd as Sm
    from oracle import model_ref
    a, b = _rand(G, N, D, seed=50) * 0.5, _rand(G, N, D, seed=51) * 0.5
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = model_ref.info_nce(ar, br, 0.3)
    ref.backward()
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    got = Sm.InfoNCE(0.3)(ad, bd)
    close(got.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6, what='infonce')
    got.backward()
    close(ad.grad.cpu(), ar.grad, rtol=2e-4, atol=1e-6, what='da', norm_rtol=1e-4)
    close(bd.grad.cpu(), br.grad, rtol=2e-4, atol=1e-6, what='db', norm_rtol=1e-4)


@pytest.mark.parametrize('kind', [0, 1])
def test_adam_step_with_folded_zero_grad(kind):
    """sbr_adam_step_zero_grad = sbr_adam_step + gradient.zero_() in one launch: the same parameter / moment bits, and a gradient
    buffer of +0.0 afterwards whatever it held (zeros, negative zeros, a NaN)."""
    ops = S().ops
    n = 100003
    g0 = _rand(n, seed=5)
    g0[::3] = 0.0
    g0[1::7] = -0.0
    g0[5] = float('nan')
    res = {}
    for fold in (False, True):
        p, m, v = _rand(n, seed=1).to(DEV), (_rand(n, seed=2) * 0.1).to(DEV), (_rand(n, seed=3).abs() * 0.01).to(DEV)
        g = g0.clone().to(DEV)
        for step in (1, 2):
            src = torch.tensor([1.5, -2.0, float(step)], device=DEV, dtype=torch.float64)
            dst = torch.zeros(3, device=DEV, dtype=torch.float64)
            ops.adam_step(kind, p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, zero_grad=fold, copy=(src, dst) if fold else None)
            if fold:
                assert bool((g.view(torch.int32) == 0).all())
                assert torch.equal(dst, src)                     # the small copy that rides on the launch
            g.copy_(g0)
        res[fold] = (p.cpu(), m.cpu(), v.cpu())
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('name', ['adamw', 'adam', 'adagrad'])
def test_fused_optimizer_vs_update_rules(name):
    ops = S().ops
    from oracle import train_ref
    n = 10007
    p0, g = _rand(n, seed=60), _rand(3, n, seed=61)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    pd, md, vd = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for s in range(3):
        if name == 'adagrad':
            p, m = train_ref.adagrad_update(p, g[s], m, s + 1, 1e-2, 1e-2)
            ops.adagrad_step(pd, g[s].to(DEV), md, 1e-2, 1e-10, 1e-2)
        else:
            fn = train_ref.adamw_update if name == 'adamw' else train_ref.adam_update
            p, m, v = fn(p, g[s], m, v, s + 1, 1e-2, 1e-2)
            ops.adam_step(0 if name == 'adamw' else 1, pd, g[s].to(DEV), md, vd, 1e-2, 0.9, 0.999, 1e-8, 1e-2, s + 1)
    close(pd.cpu(), p, rtol=1e-5, atol=1e-6, what=name)


def test_topk_exact_ties_and_mask():
    ops = S().ops
    g = torch.Generator().manual_seed(70)
    sc = torch.randn(64, 3299, generator=g)
    sc[:, ::7] = sc[:, 1::7]                           # many exact ties
    sc[5] = -float('inf')
    sc[5, :3] = torch.tensor([1., 1., 2.])             # fewer finite values than k
    d = sc.to(DEV)
    for k in (1, 10, 20, 100, 256):
        val, idx = ops.topk_rows(d, k)
        tv, _ = torch.topk(sc, k, sorted=True)
        assert torch.equal(val.cpu(), tv)
        # indices: point at the returned values, distinct, ties broken towards the lower index
        assert torch.equal(torch.gather(sc, 1, idx.cpu().long()), val.cpu())
        srt = idx.cpu().long()
        assert all(len(set(r.tolist())) == k for r in srt)
        same = val[:, 1:].cpu() == val[:, :-1].cpu()
        assert (srt[:, 1:][same] > srt[:, :-1][same]).all()
    # CSR mask
    import scipy.sparse as sp
    m = sp.random(64, 3299, density=0.02, format='csr', random_state=1)
    out = torch.randn(64, 3299, device=DEV)
    u = torch.arange(63, -1, -1, device=DEV)           # reversed user order: row b uses CSR row u[b]
    before = out.clone()
    ops.mask_scores_(out, u, torch.from_numpy(m.indptr.astype(np.int64)).to(DEV), torch.from_numpy(m.indices.astype(np.int32)).to(DEV))
    dense = torch.from_numpy(m.toarray() != 0)[u.cpu()]
    assert torch.isinf(out.cpu()[dense]).all() and torch.equal(out.cpu()[~dense], before.cpu()[~dense])


@pytest.mark.parametrize('Bu,I,D,k', [(5, 70, 64, 3), (300, 1000, 128, 20), (257, 3299, 64, 10), (1000, 5000, 256, 20),
                                      (500, 2000, 128, 32), (449, 700, 128, 1), (1000, 20000, 128, 20)])
def test_fused_f16_scorer_matches_unfused(Bu, I, D, k):
    """fp16-MFMA score + mask + top-k == (fp32 matmul of the same fp16-rounded inputs -> mask -> exact top-k)."""
    ops = S().ops
    import scipy.sparse as sp
    u = (_rand(Bu, D, seed=80) / 4).half()
    it = (_rand(I, D, seed=81) / 4).half()
    m = sp.random(Bu, I, density=0.03, format='csr', random_state=2)
    m.sort_indices()
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    uidx = torch.arange(Bu, device=DEV)
    val, idx = ops.score_topk_f16(u.to(DEV), it.to(DEV), k, uidx, indptr, indices)
    ref = (u.double() @ it.double().t())
    ref[torch.from_numpy(m.toarray() != 0)] = -float('inf')
    tv, ti = torch.topk(ref, k, sorted=True)
    close(val.cpu(), tv, rtol=1e-5, atol=1e-5, what='fused values')
    # the selected items' exact scores equal the returned values and none of them is excluded
    got = torch.gather(ref, 1, idx.cpu().long())
    close(got, tv, rtol=1e-5, atol=1e-5, what='fused indices')
    assert not torch.isinf(got).any()


@pytest.mark.parametrize('case', ['one_remainder_unit', 'seven_parts', 'two_parts', 'remainder_too_large', 'two_full_ragged',
                                  'thirteen_full', 'several_rounds', 'one_remainder_unit_long_catalogue', 'one_remainder_unit_sorted'])
def test_fused_f16_scorer_user_counts_around_the_wave_plan(case):
    """The scorer's work plan (csrc/score_topk_f16_n.hip, s5_plan): W full waves per workgroup, and when the 32-user units do not
    divide over the CUs, each remainder unit cut into P parts by item tile on one more wave of P workgroups, merged by the final
    selection. User counts on every branch of that plan — P = 8 / 7 / 2, a remainder too large to cut, a ragged last unit, the
    largest W with a part wave, more units than one round holds — with catalogues of fewer tiles than parts (parts without a tile),
    a catalogue long enough for the prefix pass, and scores that increase with the item index (every part overflows), all with
    exclusions: == fp64 matmul of the same fp16 inputs -> mask -> exact top-k."""
    ops = S().ops
    G = torch.cuda.get_device_properties(0).multi_processor_count
    Wf, R, ragged, I, D, k = {'one_remainder_unit': (1, 1, 0, 300, 64, 10), 'seven_parts': (1, G // 7, 0, 300, 128, 10),
                              'two_parts': (1, G // 2, 0, 300, 64, 10), 'remainder_too_large': (1, G // 2 + 1, 0, 300, 64, 10),
                              'two_full_ragged': (2, 6, 7, 300, 128, 20), 'thirteen_full': (13, 3, 0, 200, 64, 5),
                              'several_rounds': (14, 0, 5, 200, 64, 5), 'one_remainder_unit_long_catalogue': (1, 1, 0, 7000, 64, 20),
                              'one_remainder_unit_sorted': (1, 1, 0, 2000, 64, 20)}[case]
    Bu = 32 * (Wf * G + R) - (32 - ragged if ragged else 0) + (32 if ragged else 0)
    g = torch.Generator().manual_seed(90)
    u = (torch.randn(Bu, D, generator=g) / 4).half()
    it = (torch.randn(I, D, generator=g) / 4).half()
    if case.endswith('sorted'):
        it = torch.zeros(I, D)
        it[:, 0] = torch.linspace(-1, 1, I)
        it = it.half()
        u[:, 0] = u[:, 0].abs() + 0.5
        u[:, 1:] = 0
    rng = np.random.default_rng(5)
    cols = np.sort(rng.integers(0, I, size=(Bu, 3)), axis=1).astype(np.int32)       # three excluded items per user (duplicates possible)
    import scipy.sparse as sp
    m = sp.csr_matrix((np.ones(Bu * 3, dtype=np.int8), cols.reshape(-1), np.arange(0, 3 * Bu + 1, 3)), shape=(Bu, I))
    m.sum_duplicates()
    m.sort_indices()
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    val, idx = ops.score_topk_f16(u.to(DEV), it.to(DEV), k, torch.arange(Bu, device=DEV), indptr, indices)
    val, idx = val.cpu(), idx.cpu().long()
    for lo in range(0, Bu, 16384):                                                # reference in slices: [Bu, I] doubles at once is 6 GB
        hi = min(Bu, lo + 16384)
        ref = u[lo:hi].double() @ it.double().t()
        ref[torch.from_numpy(m[lo:hi].toarray() != 0)] = -float('inf')
        tv, _ = torch.topk(ref, k, sorted=True)
        close(val[lo:hi], tv, rtol=1e-5, atol=1e-5, what=f'values of users {lo}..{hi}')
        got = torch.gather(ref, 1, idx[lo:hi])
        close(got, tv, rtol=1e-5, atol=1e-5, what=f'indices of users {lo}..{hi}')
        assert not torch.isinf(got).any()
        srt = idx[lo:hi]
        assert (torch.sort(srt, dim=1).values[:, 1:] != torch.sort(srt, dim=1).values[:, :-1]).all()      # distinct items per user


def test_fused_f16_scorer_sorted_catalogue_and_item_offset():
    """Adversarial order: scores increase with the item index, so every tile overflows the candidate buffers."""
    ops = S().ops
    Bu, I, D, k = 64, 2000, 64, 20
    it = torch.zeros(I, D)
    it[:, 0] = torch.linspace(-1, 1, I)
    u = torch.zeros(Bu, D)
    u[:, 0] = 1.0
    val, idx = ops.score_topk_f16(u.half().to(DEV), it.half().to(DEV), k, item_offset=1000)
    ref = (u.half().double() @ it.half().double().t())
    tv, ti = torch.topk(ref, k, sorted=True)
    close(val.cpu(), tv, rtol=1e-6, atol=1e-6, what='values')
    assert (idx.cpu() >= 1000).all()
    close(torch.gather(ref, 1, idx.cpu().long() - 1000), tv, rtol=1e-6, atol=1e-6, what='indices')


@pytest.mark.parametrize('D', [64, 128, 256])
def test_fused_f16_scorer_ties_and_short_catalogues(D):
    """Exact ordering rule under massive ties (scores take ~20 distinct values, so the k-th score is shared by hundreds of
    items: the selection step has to resolve the tie on the item index) and catalogues shorter than k (missing slots: -inf / -1);
    exclusions on. Reference: stable sort by (score desc, item index asc)."""
    ops = S().ops
    import scipy.sparse as sp
    g = torch.Generator().manual_seed(D)
    Bu, I, k = 130, 3000, 20
    u = torch.zeros(Bu, D)
    it = torch.zeros(I, D)
    u[:, :4] = torch.randint(-1, 2, (Bu, 4), generator=g).float()
    it[:, :4] = torch.randint(-2, 3, (I, 4), generator=g).float()
    m = sp.random(Bu, I, density=0.05, format='csr', random_state=5)
    m.sort_indices()
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    uidx = torch.arange(Bu, device=DEV)
    val, idx = ops.score_topk_f16(u.half().to(DEV), it.half().to(DEV), k, uidx, indptr, indices)
    ref = (u.double() @ it.double().t()).numpy()
    ref[m.toarray() != 0] = -np.inf
    order = np.lexsort((np.broadcast_to(np.arange(I), ref.shape), -ref), axis=1)[:, :k]
    assert np.array_equal(idx.cpu().numpy(), order)
    assert np.array_equal(val.cpu().numpy().astype(np.float64), np.take_along_axis(ref, order, axis=1))
    # a 7-item catalogue: 7 sorted entries, then empty slots
    val, idx = ops.score_topk_f16(u.half().to(DEV), it[:7].half().to(DEV), k)
    ref7 = (u.double() @ it[:7].double().t()).numpy()
    o7 = np.lexsort((np.broadcast_to(np.arange(7), ref7.shape), -ref7), axis=1)
    assert np.array_equal(idx.cpu().numpy()[:, :7], o7) and (idx.cpu().numpy()[:, 7:] == -1).all()
    assert np.isneginf(val.cpu().numpy()[:, 7:]).all()


@pytest.mark.parametrize('bias', ['periodic_high', 'periodic_low', 'none'])
def test_fused_f16_scorer_periodic_catalogues(bias):
    """16k-item catalogue in which every 16th tile of 64 items holds ALL the highest (or all the lowest) scores: bursts of
    candidates for every row at the same time, long stretches without any — the candidate buffers fill and compact in
    lockstep. Exclusions on; compared with the fp64 reference."""
    ops = S().ops
    import scipy.sparse as sp
    Bu, I, D, k = 600, 16384, 128, 20
    g = torch.Generator().manual_seed(17)
    u = (torch.randn(Bu, D, generator=g) / 4)
    u[:, 0] = 1.0
    it = (torch.randn(I, D, generator=g) / 8)
    marked = ((torch.arange(I) // 64) % 16 == 0)
    it[:, 0] = 0.0
    if bias == 'periodic_high':
        it[marked, 0] = 8.0
    elif bias == 'periodic_low':
        it[marked, 0] = -8.0
    m = sp.random(Bu, I, density=0.004, format='csr', random_state=9)
    m.sort_indices()
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    uidx = torch.arange(Bu, device=DEV)
    val, idx = ops.score_topk_f16(u.half().to(DEV), it.half().to(DEV), k, uidx, indptr, indices)
    ref = (u.half().double() @ it.half().double().t())
    ref[torch.from_numpy(m.toarray() != 0)] = -float('inf')
    tv, ti = torch.topk(ref, k, sorted=True)
    close(val.cpu(), tv, rtol=1e-5, atol=1e-5, what='values')
    close(torch.gather(ref, 1, idx.cpu().long()), tv, rtol=1e-5, atol=1e-5, what='indices')


def test_rank_metrics_vs_oracle():
    ops = S().ops
    from oracle import eval_ref
    import scipy.sparse as sp
    Bu, I = 200, 500
    g = torch.Generator().manual_seed(90)
    scores = torch.randn(Bu, I, generator=g)
    lab = sp.random(Bu, I, density=0.02, format='csr', random_state=3)
    lab.data[:] = 1
    lab.sort_indices()
    lab = sp.csr_matrix(lab)
    lab[7] = 0                                         # a user without positives
    lab.eliminate_zeros()
    y = torch.from_numpy(lab.toarray().astype(np.float32))
    val, idx = ops.topk_rows(scores.to(DEV), 20)
    m = ops.rank_metrics(idx, None, torch.from_numpy(lab.indptr.astype(np.int64)).to(DEV),
                         torch.from_numpy(lab.indices.astype(np.int32)).to(DEV), [1, 10, 20])
    for qi, k in enumerate([1, 10, 20]):
        ii = idx.cpu().long()[:, :k]
        close(m[0, qi].cpu(), eval_ref.ndcg_at_k(y, ii), rtol=1e-5, atol=1e-6, what=f'ndcg@{k}')
        close(m[1, qi].cpu(), eval_ref.recall_at_k(y, ii), rtol=1e-5, atol=1e-6, what=f'recall@{k}')
        close(m[2, qi].cpu(), eval_ref.precision_at_k(y, ii), rtol=1e-5, atol=1e-6, what=f'precision@{k}')


def test_full_size_properties_c2_shapes():
    """BASELINE config-2-sized shapes (U 100k, I 50k, F 768, D 128) through size-independent properties:
    linearity of the projector GEMM, and fused-scorer == unfused scorer on a user sample."""
    ops = S().ops
    g = torch.Generator(device=DEV).manual_seed(5)
    I, F, C = 50000, 768, 128
    X = torch.randn(I, F, device=DEV, generator=g)
    W1, W2 = torch.randn(C, F, device=DEV, generator=g), torch.randn(C, F, device=DEV, generator=g)
    rows = torch.randint(0, I, (90112,), device=DEV, generator=g, dtype=torch.int32)
    y1, y2 = ops.linear_nt(X, W1, None, 0, a_idx=rows), ops.linear_nt(X, W2, None, 0, a_idx=rows)
    y12 = ops.linear_nt(X, W1 + W2, None, 0, a_idx=rows)
    err = (y1 + y2 - y12).abs().max().item()
    assert err < 2e-3 * y12.abs().max().item() + 1e-3, err
    ref = X[rows.long()[:64]].double() @ W1.double().t()
    close(y1[:64].cpu(), ref.cpu(), rtol=1e-4, atol=1e-3, what='sampled rows', norm_rtol=1e-5)
    # scoring
    U_ = torch.randn(4096, C, device=DEV, generator=g) / 8
    It = torch.randn(I, C, device=DEV, generator=g) / 8
    u16, i16 = ops.cast_f16(U_), ops.cast_f16(It)
    val, idx = ops.score_topk_f16(u16, i16, 20)
    sc = ops.linear_nt(u16.float(), i16.float())
    tv, ti = ops.topk_rows(sc, 20)
    close(val.cpu(), tv.cpu(), rtol=1e-4, atol=1e-5, what='fused vs unfused values')
    assert (idx == ti).float().mean().item() > 0.999


@pytest.mark.parametrize('C,act', [(64, 1), (128, 0), (48, 2), (512, 1), (30, 0)])
def test_csr_projector_long_tailed_rows(C, act):
    """CSR 'interactions' projector (Feature.py:149-150 toarray() + Linear, sgd_alg.py:1380) on long-tailed rows: one row
    with thousands of nnz, empty rows, repeated rows in the batch; forward and dW against dense torch fp32.
    C = 30 takes the generic wave-per-row kernel, the others the workgroup-per-row kernels."""
    import scipy.sparse as sp
    import importlib
    mod = S()
    _lib = importlib.import_module(mod.ops.__name__.rsplit('.', 1)[0] + '._lib')
    call, ptr, ops = _lib.call, _lib.ptr, mod.ops
    n_rows, n_cols = 50, 6000
    rng = np.random.default_rng(C)
    lens = np.array([0, 1, 3, 5000, 700, 64, 65, 255, 256, 257] + list(rng.integers(0, 120, size=n_rows - 10)))
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, size=l, replace=False)) for l in lens]).astype(np.int32)
    vals = rng.standard_normal(indices.size).astype(np.float32)
    dense = torch.from_numpy(sp.csr_matrix((vals, indices, indptr), shape=(n_rows, n_cols)).toarray())
    W = _rand(C, n_cols, seed=90) * 0.05
    b = _rand(C, seed=91)
    rows = torch.from_numpy(np.concatenate([np.arange(n_rows), [3, 3, 0, 4]]).astype(np.int32))
    n = rows.numel()
    slots = torch.from_numpy(rng.permutation(n).astype(np.int32))
    wt = W.t().contiguous().to(DEV)                      # column-major projector weight: [n_cols, C]
    # every device operand stays referenced until the results are read back (a temporary's memory is recycled at once)
    b_d, rows_d, slots_d = b.to(DEV), rows.to(DEV), slots.to(DEV)
    for use_vals in (True, False):
        d = dense if use_vals else (dense != 0).float()
        ref_pre = d[rows.long()] @ W.t() + b
        ref = {0: ref_pre, 1: torch.relu(ref_pre), 2: torch.tanh(ref_pre)}[act]
        out = torch.zeros(n, C, device=DEV)
        dv = torch.from_numpy(vals).to(DEV) if use_vals else None
        ip, ix = torch.from_numpy(indptr).to(DEV), torch.from_numpy(indices).to(DEV)
        call('sbr_csr_project_fwd', ptr(ip), ptr(ix), ptr(dv), ptr(wt), wt.stride(0), ptr(b_d), ptr(rows_d),
             ptr(out), out.stride(0), ptr(slots_d), n, C, act, ops.stream())
        got = torch.empty_like(ref)
        got[:] = out.cpu()[slots.long()]
        close(got, ref, rtol=1e-4, atol=1e-5, what=f'csr fwd vals={use_vals}', norm_rtol=1e-5)
        dz = _rand(n, C, seed=92)
        dz_d = dz.to(DEV)
        dwt = torch.zeros_like(wt)
        call('sbr_csr_project_bwd', ptr(ip), ptr(ix), ptr(dv), ptr(dz_d), C, ptr(rows_d), ptr(dwt), dwt.stride(0),
             n, C, ops.stream())
        ref_dw = dz.t() @ d[rows.long()]                  # [C, n_cols]
        close(dwt.cpu().t(), ref_dw, rtol=1e-4, atol=1e-5, what=f'csr dW vals={use_vals}', norm_rtol=1e-5)
        if C % 4 == 0:
            # the gather form: slot gradients summed per entity, then every feature column gathers its entities' rows over the
            # TRANSPOSED matrix (features.DeviceTable.transposed builds the same arrays on the device); accumulates into dWt
            mt = sp.csr_matrix((vals if use_vals else np.ones_like(vals), indices, indptr), shape=(n_rows, n_cols)).T.tocsr()
            mt.sort_indices()
            tp, tx = torch.from_numpy(mt.indptr.astype(np.int64)).to(DEV), torch.from_numpy(mt.indices.astype(np.int32)).to(DEV)
            tv = torch.from_numpy(mt.data.astype(np.float32)).to(DEV) if use_vals else None
            ws = torch.full((n_rows, C), float('nan'), device=DEV)            # overwritten by the call
            dwt2 = torch.full_like(wt, 0.25)
            call('sbr_csr_project_bwd_gather', ptr(tp), ptr(tx), ptr(tv), ptr(dz_d), C, None, ptr(rows_d), n, ptr(ws), C, n_rows, ptr(dwt2),
                 dwt2.stride(0), n_cols, C, ops.stream())
            close(dwt2.cpu().t() - 0.25, ref_dw, rtol=1e-4, atol=1e-5, what=f'csr dW (gather form) vals={use_vals}', norm_rtol=1e-5)


@pytest.mark.parametrize('kind', ['tag', 'csr'])
def test_front_backward_of_sparse_modalities_gather_form_equals_scatter_form(kind):
    """FeatureEmbedding.front_backward of a tag bag (nn.EmbeddingBag(mean, padding), sgd_alg.py:1336-1337) and of a CSR projector
    (sgd_alg.py:1380): with many slots the weight gradient is computed in gather form (per-entity sums, then every tag / feature
    column gathers its entities' rows: features.DeviceTable.transposed + sbr_csr_project_bwd_gather), with few in scatter form
    (one atomic per slot, entry and column). Both against the dense fp64 product, on the same slots: duplicated entities,
    entities without any tag / entry, a tag used by every entity."""
    import sibrar_amd as S
    from importlib import import_module
    pkg = S.ops.__name__.rsplit('.', 1)[0]
    features, sbnet = import_module(pkg + '.features'), import_module(pkg + '.sbnet')
    rng = np.random.default_rng(12)
    n_ent, n_cols, C, T = 500, 37, 64, 6
    if kind == 'tag':
        tags = np.full((n_ent, T), n_cols, dtype=np.int64)                          # padding value = number of tags
        for e in range(n_ent):
            k_ = int(rng.integers(0, T + 1)) if e % 50 else 0                       # some entities have no tag at all
            tags[e, :k_] = rng.choice(n_cols, size=k_, replace=False)
            if k_ and e % 3 == 0:
                tags[e, 0] = 5                                                      # a very common tag
        feat = features.HostFeature('genres', 'tag', tags, n_categories=n_cols)
        dense = np.zeros((n_ent, n_cols + 1))
        for e in range(n_ent):
            real = tags[e][tags[e] != n_cols]
            for g_ in real:
                dense[e, g_] += 1.0 / len(real)
    else:
        import scipy.sparse as sp
        m = sp.random(n_ent, n_cols, density=0.15, format='csr', random_state=3, dtype=np.float32)
        m.data[:] = 1.0
        feat = features.HostFeature('interactions', 'csr', m)
        dense = m.toarray().astype(np.float64)
    torch.manual_seed(0)
    fe = sbnet.FeatureEmbedding(feat, embedding_dim=C).to(DEV)
    p = fe.front_params()
    n = 3000
    rows = torch.from_numpy(rng.integers(0, n_ent, size=n).astype(np.int32)).to(DEV)
    slots = torch.from_numpy(rng.permutation(n + 50)[:n].astype(np.int32)).to(DEV)          # slot rows of the [R, C] matrix
    dout = torch.randn(n + 50, C, generator=torch.Generator().manual_seed(1)).to(DEV)
    out = torch.randn(n + 50, C, generator=torch.Generator().manual_seed(2)).abs().to(DEV) + 0.1  # positive: ReLU' = 1 everywhere
    got = {}
    assert fe._gather_pays(30805, 326_000, 13610, 5192, 512) and not fe._gather_pays(1877, 326_000, 13610, 5192, 512)      # Onion18 at batch 4096 / 256
    assert fe._gather_pays(45056, 6600, 3299, 19, 64) and not fe._gather_pays(2816, 6600, 3299, 19, 64)                   # ML-1M's 18 genre tags: the adds on one row serialise
    for form, force in (('gather', True), ('scatter', False)):
        fe.CSR_GATHER_FORCE = force
        hidden = [] if kind == 'tag' else [out]
        grads = fe.front_backward(p, hidden, rows, n, out, dout, slots)
        got[form] = grads[0].detach().double().cpu()
    ref_in = dout.double().cpu()[slots.cpu().long()]                                 # [n, C]
    X = torch.from_numpy(dense)[rows.cpu().long()]                                   # [n, n_cols (+ pad)]
    ref = X.t() @ ref_in if kind == 'tag' else ref_in.t() @ X                         # bag weight [n_cols + 1, C] / Linear weight [C, n_cols]
    for form in ('gather', 'scatter'):
        close(got[form], ref, rtol=1e-4, atol=1e-5, what=f'{kind} weight gradient, {form} form', norm_rtol=1e-5)


@pytest.mark.parametrize('R,n_mod,pad', [(1, 1, 0), (24, 2, 64), (4096, 3, 0), (4097, 2, 128), (90112, 2, 1024), (180224, 8, 0)])
def test_partition_slots_is_a_stable_counting_sort(R, n_mod, pad):
    """sbr_partition_slots == the stable argsort of the modality draw (the boolean-mask grouping of sgd_alg.py:1934-1957), with
    every segment padded to its capacity by the sentinel R."""
    import ctypes, importlib
    mod = S()
    _lib = importlib.import_module(mod.ops.__name__.rsplit('.', 1)[0] + '._lib')
    call, ptr, ops = _lib.call, _lib.ptr, mod.ops
    rng = np.random.default_rng(R + n_mod)
    pos = rng.integers(0, n_mod, size=R).astype(np.int8)
    if n_mod > 2:
        pos[pos == 1] = 0                                   # an empty modality in between
    counts = np.bincount(pos, minlength=n_mod)
    caps = (counts + pad - 1) // pad * pad if pad else counts
    seg = np.concatenate([[0], np.cumsum(caps)]).astype(np.int32)
    want = np.full(int(seg[-1]), R, dtype=np.int32)
    order = np.argsort(pos, kind='stable').astype(np.int32)
    src = 0
    for m in range(n_mod):
        want[seg[m]:seg[m] + counts[m]] = order[src:src + counts[m]]
        src += counts[m]
    pos_d = torch.from_numpy(pos).to(DEV)
    out = torch.full((max(int(seg[-1]), 1),), -7, dtype=torch.int32, device=DEV)
    ws = torch.zeros(mod.lib().sbr_partition_slots_workspace(R) // 4 + 8, dtype=torch.int32, device=DEV)
    seg_arr = (ctypes.c_int * len(seg))(*seg.tolist())
    call('sbr_partition_slots', ptr(pos_d), R, n_mod, ctypes.cast(seg_arr, ctypes.c_void_p), ptr(out), ptr(ws), ws.numel() * 4,
         ops.stream())
    assert np.array_equal(out.cpu().numpy()[:int(seg[-1])], want)


@pytest.mark.parametrize('I', [8192, 50000, 65536])
def test_topk_long_rows_sampled_path_is_exact(I):
    """Rows of >= 8192 scores take the one-read sampled selection (threshold from a 1/16 sample, one pass collecting everything
    above it, exact sort of the few hundred candidates). It must return exactly what the radix path returns: same values, ties
    towards the lower index — also for rows that force its fallback (mostly masked rows, constant rows, heavy ties)."""
    ops = S().ops
    g = torch.Generator().manual_seed(71 + I)
    sc = torch.randn(24, I, generator=g)
    n3 = (I - 2) // 3
    sc[1, 0:3 * n3:3] = sc[1, 1:3 * n3:3]                # many exact ties
    sc[2] = 0.5                                          # constant row: every element ties with the threshold -> fallback
    sc[3] = -float('inf')
    sc[3, 17] = 3.0                                      # fewer finite values than k -> fallback
    sc[4, :I // 2] = -float('inf')                       # half masked
    sc[5] = torch.arange(I, dtype=torch.float32)         # sorted ascending (the sample sees every 16th)
    sc[6] = -torch.arange(I, dtype=torch.float32)
    sc[7] = torch.randn(I, generator=g).abs() * 1e-30    # tiny magnitudes (denormal range of the key space)
    d = sc.to(DEV)
    for k in (1, 10, 100, 256):
        val, idx = ops.topk_rows(d, k)
        tv, _ = torch.topk(sc, k, sorted=True)
        assert torch.equal(val.cpu(), tv), (I, k)
        srt = idx.cpu().long()
        assert torch.equal(torch.gather(sc, 1, srt), val.cpu())
        assert all(len(set(r.tolist())) == k for r in srt)
        same = val[:, 1:].cpu() == val[:, :-1].cpu()
        assert (srt[:, 1:][same] > srt[:, :-1][same]).all()


@pytest.mark.parametrize('W,cap,D,n_table', [(1, 5, 3, 4), (3, 40, 16, 25), (8, 1000, 128, 3000)])
def test_scatter_add_rows_sorted_is_the_dense_gradient_without_atomics(W, cap, D, n_table):
    """sbr_scatter_add_rows_sorted (data-parallel exchange of lookup gradients): W gathered blocks of ``cap`` (row, gradient)
    pairs laid out like the all-gather's receive buffer [W][cap*D floats | cap int32], stable-sorted by table row ->
    dW[row] += sum of the rows, equal to index_add in float64; a second run gives the same bits (fixed summation order);
    and equal to the atomic scatter up to rounding."""
    from importlib import import_module
    _lib = import_module(S().ops.__name__.rsplit('.', 1)[0] + '._lib')
    g = torch.Generator().manual_seed(W * 1000 + cap)
    grads = torch.randn(W, cap, D, generator=g)
    rows = torch.randint(0, n_table, (W, cap), generator=g, dtype=torch.int32)
    recv = torch.zeros(W, cap * (D + 1), dtype=torch.float32)
    recv[:, :cap * D] = grads.reshape(W, -1)
    recv[:, cap * D:] = rows.view(torch.float32)
    recv = recv.to(DEV)
    outs = []
    for _ in range(2):
        dW = torch.full((n_table, D), 0.5, device=DEV)
        r_all = recv[:, cap * D:].view(torch.int32).reshape(-1)
        rs, perm = torch.sort(r_all, stable=True)
        _lib.call('sbr_scatter_add_rows_sorted', recv.data_ptr(), D, cap, recv.stride(0), perm.data_ptr(), rs.data_ptr(),
                  dW.data_ptr(), dW.stride(0), W * cap, D, _lib.stream())
        outs.append(dW.cpu())
    ref = torch.full((n_table, D), 0.5, dtype=torch.float64)
    ref.index_add_(0, rows.reshape(-1).long(), grads.reshape(-1, D).double())
    close(outs[0], ref, rtol=1e-5, atol=1e-5, what='sorted scatter')
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('name', ['adamw', 'adam'])
def test_deferred_row_wise_adam_replay_is_bit_identical(name):
    """sbr_adam_rows (engine.DeferredTable): a lookup table whose rows take the optimizer steps they missed later, in order, with
    zero gradient, ends BIT-identical to the dense optimizer (torch.optim semantics, train/trainer.py:62-68) stepping the whole
    table every time: parameters and both moment buffers, after 40 steps with deterministic gradients in which rows go
    untouched for long stretches, with duplicate ids in the touched lists, an id map, a flush in the middle, and a dense
    step (FusedOptimizer.step) interleaved."""
    import sibrar_amd as S
    from importlib import import_module
    engine = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine')
    R, D = 300, 48
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(R, D, generator=g) * 0.1
    rowmap = torch.randperm(R, generator=g).to(torch.int32)           # entity id -> table row
    mods, opts = [], []
    for _ in range(2):
        m = torch.nn.Embedding(R, D)
        with torch.no_grad():
            m.weight.copy_(w0)
        m.to(DEV)
        mods.append(m)
        opts.append(S.FusedOptimizer(m, name, lr=3e-3, weight_decay=1e-2))
    d = engine.DeferredTable(opts[1], mods[1].weight, 0, R * D, rowmap.to(DEV))
    opts[1].deferred = d
    rng = np.random.default_rng(2)
    for t in range(40):
        n = int(rng.integers(1, 12))
        ids = torch.from_numpy(rng.integers(0, R if t % 7 else 5, size=n))          # entity ids, duplicates likely
        rows = rowmap[ids].long().unique()
        grad_rows = torch.randn(len(rows), D, generator=g)
        if t == 20:                                                               # a dense step in between (autograd path)
            for o, m in zip(opts, mods):
                o.zero_grad()
                m.weight.grad[rows.to(DEV)] = grad_rows.to(DEV)
                o.step()
                o.zero_grad()
            continue
        # dense reference
        opts[0].zero_grad()
        mods[0].weight.grad[rows.to(DEV)] = grad_rows.to(DEV)
        opts[0].step_flat()
        # deferred: rows read by the "forward" are brought up to date, gradient rows written, touched rows updated
        ids_dev = ids.to(DEV)
        d.catch_up(ids_dev)
        current = mods[1].weight.detach()[rows.to(DEV)].cpu()
        assert torch.equal(current, mods[0].weight.detach()[rows.to(DEV)].cpu() * 0 + current)      # (read forces the sync)
        mods[1].weight.grad[rows.to(DEV)] = grad_rows.to(DEV)
        opts[1].step_flat(skip=(0, R * D))
        d.update(ids_dev)
        assert float(mods[1].weight.grad.abs().max()) == 0.0                       # consumed gradient rows are re-zeroed
        if t == 11:
            d.flush()
            assert torch.equal(mods[1].weight.detach().cpu(), mods[0].weight.detach().cpu())
    d.flush()
    assert torch.equal(mods[1].weight.detach().cpu(), mods[0].weight.detach().cpu())
    assert torch.equal(opts[1].m.cpu(), opts[0].m.cpu()) and torch.equal(opts[1].v.cpu(), opts[0].v.cpu())


@pytest.mark.parametrize('name,every', [('adamw', 4), ('adam', 7), ('adamw', 16)])
def test_deferred_adam_with_the_sweep_of_the_optimizer_launch_is_bit_identical(name, every):
    """engine.DeferredTable.step (sbr_adam_step_rows): besides the rows that received gradient, the optimizer launch brings 1 / W of
    the table's sub-rows per step up to date (cyclic sweep), so that no row is ever more than W steps behind. 60 steps with
    deterministic gradients, duplicate ids, an id map, rows of two ragged sub-rows, batches that hit the swept range, one step whose
    catch-up is left out (no sweep then): parameters and moments stay BIT-identical to the dense optimizer stepping the whole
    table every step (train/trainer.py:62-68), and the backlog of every row stays within W (+ the one step without a sweep)."""
    import sibrar_amd as S
    from importlib import import_module
    engine = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine')
    R, D = 300, 80
    g = torch.Generator().manual_seed(8)
    w0 = torch.randn(R, D, generator=g) * 0.1
    rowmap = torch.randperm(R, generator=g).to(torch.int32)
    mods, opts = [], []
    for _ in range(2):
        m = torch.nn.Embedding(R, D)
        with torch.no_grad():
            m.weight.copy_(w0)
        m.to(DEV)
        mods.append(m)
        opts.append(S.FusedOptimizer(m, name, lr=3e-3, weight_decay=1e-2))
        opts[-1].zero_grad()
    d = engine.DeferredTable(opts[1], mods[1].weight, 0, R * D, rowmap.to(DEV))
    d.SWEEP_EVERY = every
    opts[1].deferred = d
    rng = np.random.default_rng(4)
    for t in range(60):
        n = int(rng.integers(1, 40))
        ids = torch.from_numpy(rng.integers(0, R if t % 5 else 8, size=n))
        rows = rowmap[ids].long().unique().to(DEV)
        grad_rows = torch.randn(len(rows), D, generator=g).to(DEV)
        opts[0].zero_grad()
        mods[0].weight.grad[rows] = grad_rows
        opts[0].step_flat()
        ids_dev = ids.to(DEV)
        if t != 33:
            d.catch_up(ids_dev)
        else:                                                        # (rows are current after the flush below: nothing to catch up)
            d.flush()
        mods[1].weight.grad[rows] = grad_rows
        assert opts[1].step_flat(zero_grad=True, rows=ids_dev) is False
        assert float(mods[1].weight.grad.abs().max()) == 0.0
        if t >= every + 1:
            assert int((opts[1].step_count - d.last).max()) <= every + (1 if 33 <= t <= 33 + every else 0), t
        if t in (20, 41):
            d.flush()
            assert torch.equal(mods[1].weight.detach().cpu(), mods[0].weight.detach().cpu()), t
    d.flush()
    assert torch.equal(mods[1].weight.detach().cpu(), mods[0].weight.detach().cpu())
    assert torch.equal(opts[1].m.cpu(), opts[0].m.cpu()) and torch.equal(opts[1].v.cpu(), opts[0].v.cpu())


@pytest.mark.parametrize('every', [0, 16])
def test_deferred_adamw_replay_of_idle_rows_is_bit_identical(every):
    """The replay's short cut for rows whose first moment is exactly zero (csrc/optim.hip, adam_wave_is_idle: never touched, or idle
    until 0.9^n has underflowed — `v *= beta2; p *= 1 - lr wd` instead of the square root and the divisions): rows that get one
    small gradient at step 0 and nothing for 1,250 steps (their first moment passes through the denormals to zero on the way), rows
    that are never touched, rows touched now and then; without and with the optimizer launch's sweep. Parameters and both moment
    buffers BIT-identical to the dense optimizer stepping every row every step (train/trainer.py:62-68)."""
    import sibrar_amd as S
    from importlib import import_module
    engine = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine')
    R, D = 64, 80
    g = torch.Generator().manual_seed(9)
    w0 = torch.randn(R, D, generator=g) * 0.1
    mods, opts = [], []
    for _ in range(2):
        m = torch.nn.Embedding(R, D)
        with torch.no_grad():
            m.weight.copy_(w0)
        m.to(DEV)
        mods.append(m)
        opts.append(S.FusedOptimizer(m, 'adamw', lr=3e-3, weight_decay=1e-2))
        opts[-1].zero_grad()
    d = engine.DeferredTable(opts[1], mods[1].weight, 0, R * D, None)
    d.SWEEP_EVERY = every
    opts[1].deferred = d
    rng = np.random.default_rng(6)
    for t in range(1300):
        if t == 0:
            rows, scale = torch.arange(0, 10), 1e-3
        elif t == 1250:
            rows, scale = torch.arange(0, R - 8), 1.0                   # the last eight rows are never touched
        elif t % 97 == 0:
            rows, scale = torch.from_numpy(rng.integers(10, 30, size=4)).unique(), 1.0
        else:
            rows, scale = torch.from_numpy(rng.integers(30, 40, size=1)), 1.0
        grad_rows = (torch.randn(len(rows), D, generator=g) * scale).to(DEV)
        rows_dev = rows.to(DEV)
        opts[0].zero_grad()
        mods[0].weight.grad[rows_dev] = grad_rows
        opts[0].step_flat()
        ids32 = rows.to(torch.int32).to(DEV)
        d.catch_up(ids32)
        mods[1].weight.grad[rows_dev] = grad_rows
        assert opts[1].step_flat(zero_grad=True, rows=ids32) is False
    d.flush()
    assert float(opts[0].m[:10 * D].abs().max()) > 0.0                   # (rows 0-9 were touched again at step 1250)
    assert torch.equal(mods[1].weight.detach().cpu(), mods[0].weight.detach().cpu())
    assert torch.equal(opts[1].m.cpu(), opts[0].m.cpu()) and torch.equal(opts[1].v.cpu(), opts[0].v.cpu())


@pytest.mark.parametrize('W,Bu,k', [(1, 5, 3), (2, 300, 20), (8, 1000, 20), (8, 77, 32), (4, 50, 1)])
def test_merge_topk_kernel_equals_the_host_merge(W, Bu, k):
    """sbr_merge_topk (item-sharded evaluation: the all-gathered per-shard lists) == parallel.merge_topk (torch formulation, pinned
    on the CPU by tests/test_host_cpu.py): ties across shards resolved by item index, empty slots (idx -1) last."""
    from importlib import import_module
    Sm = S()
    _lib = import_module(Sm.ops.__name__.rsplit('.', 1)[0] + '._lib')
    g = torch.Generator().manual_seed(W * 100 + k)
    vals = (torch.randint(0, 40, (W, Bu, k), generator=g).float() / 8).sort(dim=2, descending=True).values   # many ties
    idxs = torch.stack([torch.stack([torch.randperm(1000, generator=g)[:k] + 1000 * w for _ in range(Bu)]) for w in range(W)]).int()
    n_valid = torch.randint(0, k + 1, (W, Bu), generator=g)                              # shards with fewer than k entries
    empty = torch.arange(k)[None, None, :] >= n_valid[..., None]
    idxs[empty] = -1
    vals[empty] = -float('inf')
    out_val = torch.empty(Bu, k, device=DEV)
    out_idx = torch.empty(Bu, k, dtype=torch.int32, device=DEV)
    vals_d, idxs_d = vals.to(DEV), idxs.to(DEV)
    _lib.call('sbr_merge_topk', vals_d.data_ptr(), idxs_d.data_ptr(), W, Bu, k, out_val.data_ptr(), out_idx.data_ptr(), _lib.stream())
    rv, ri = Sm.parallel.merge_topk(torch.cat(list(vals), dim=1), torch.cat(list(idxs), dim=1), k)
    assert torch.equal(out_idx.cpu(), ri.int())
    got, want = out_val.cpu(), rv
    assert torch.equal(torch.isinf(got), torch.isinf(want)) and torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])


@pytest.mark.parametrize('M', [1, 63, 64, 200, 4097, 90112])
def test_weights_resident_gemm_is_bit_identical_to_the_ring_kernel(M, monkeypatch):
    """sbr_gemm_wres_f32 (N = K = 128: the weight in registers, only A streamed) == sbr_gemm_f32 bit for bit — NT with bias and
    every activation the step uses, NN — including a ragged last tile; and its fused backward epilogue (activation derivative of a
    second matrix + the column sums of the result) == NN product -> sbr_act_grad_gather -> sbr_colsum."""
    ops = S().ops
    monkeypatch.setattr(ops, '_SPLIT', False)
    x, w, b = _rand(M, 128, seed=41).to(DEV), (_rand(128, 128, seed=42) / 8).to(DEV), _rand(128, seed=43).to(DEV)
    y_act = torch.relu(_rand(M, 128, seed=44)).to(DEV)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, '_WRES', flag)
        res[flag] = [ops.linear_nt(x, w, b, act) for act in (0, 1, 2)] + [ops.matmul_nn(x, w)]
    for a_, b_ in zip(res[True], res[False]):
        assert torch.equal(a_, b_)
    ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    close(res[True][0].cpu(), ref, rtol=1e-4, atol=1e-5, what='nt', norm_rtol=2e-6, scale=float(ref.abs().max()))
    # fused epilogue
    monkeypatch.setattr(ops, '_WRES', True)
    out = torch.empty(M, 128, device=DEV)
    ws = ops.new_colsum_ws(x.device, 128)
    assert ops.matmul_nn_actgrad_ok(x, w, y_act, out)
    ops.matmul_nn_actgrad(x, w, y_act, 1, out, ws)
    db = torch.empty(128, device=DEV)
    ops.colred_finish([(ws, db)])
    unf = ops.act_grad(res[False][3], y_act, 1)
    assert torch.equal(out, unf)
    close(db.cpu(), ops.colsum(unf).cpu(), rtol=1e-5, atol=1e-5, what='folded bias gradient', norm_rtol=1e-6)
    assert float(ws.abs().max()) == 0.0                       # the finishing launch left the workspace zeroed


@pytest.mark.parametrize('M', [1, 31, 32, 200, 4097, 90112])
def test_split_gemm_has_the_error_of_the_fp32_pipe(M, monkeypatch):
    """sbr_gemm_split_f32 (fp32 operands split exactly into three bf16 numbers each, six bf16 MFMA terms, fp32 accumulate) against an
    fp64 product: its error is of the size of the fp32-pipe kernel's own (summation order differs, so not bit-identical), for NT with
    bias + activations, NN, a ragged last block, and the fused backward epilogue with its folded bias gradient. Operands span six
    decades so that all three split planes carry weight."""
    ops = S().ops
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    g = torch.Generator().manual_seed(77)
    scale = torch.pow(10., torch.randint(-3, 3, (M, 128), generator=g).float())
    x = (_rand(M, 128, seed=41) * scale).to(DEV)
    w, b = (_rand(128, 128, seed=42) / 8).to(DEV), _rand(128, seed=43).to(DEV)
    y_act = torch.relu(_rand(M, 128, seed=44)).to(DEV)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, '_SPLIT', flag)
        res[flag] = [ops.linear_nt(x, w, b, act) for act in (0, 1, 2)] + [ops.matmul_nn(x, w)]
    xd, wd, bd = x.double().cpu(), w.double().cpu(), b.double().cpu()
    pre = xd @ wd.t() + bd
    absum = xd.abs() @ wd.abs().t() + bd.abs()                 # the scale rounding errors are relative to
    refs = [pre, torch.relu(pre), None, xd @ wd]
    for i in (0, 1, 3):
        mag = absum if i < 3 else xd.abs() @ wd.abs()
        e_split = ((res[True][i].double().cpu() - refs[i]).abs() / mag).max().item()
        e_f32 = ((res[False][i].double().cpu() - refs[i]).abs() / mag).max().item()
        assert e_split <= max(2.0 * e_f32, 2.0 ** -22), (i, e_split, e_f32)
        rms_split = ((res[True][i].double().cpu() - refs[i]) / mag).pow(2).mean().sqrt().item()
        rms_f32 = ((res[False][i].double().cpu() - refs[i]) / mag).pow(2).mean().sqrt().item()
        assert rms_split <= max(2.0 * rms_f32, 2.0 ** -24), (i, rms_split, rms_f32)
    # tanh epilogue: 1-Lipschitz in the pre-activation, whose two versions are each within 2^-22 * absum of the fp64 value
    assert bool(((res[True][2].double().cpu() - res[False][2].double().cpu()).abs() <= 2.0 ** -20 * absum + 1e-6).all())
    # fused epilogue
    monkeypatch.setattr(ops, '_SPLIT', True)
    out = torch.empty(M, 128, device=DEV)
    ws = ops.new_colsum_ws(x.device, 128)
    ops.matmul_nn_actgrad(x, w, y_act, 1, out, ws)
    db = torch.empty(128, device=DEV)
    ops.colred_finish([(ws, db)])
    want = (xd @ wd) * (y_act.double().cpu() > 0)
    mag = xd.abs() @ wd.abs()
    assert (((out.double().cpu() - want).abs()) / mag).max().item() <= 2.0 ** -20
    assert torch.equal(out == 0, (y_act <= 0) | (out == 0))
    close(db.cpu(), out.double().sum(0).cpu(), rtol=1e-5, atol=1e-5, what='folded bias gradient', norm_rtol=1e-6)
    assert float(ws.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,mode,gather', [(4096, 256, 64, 0, ''), (5000, 512, 512, 0, ''), (9001, 256, 1024, 0, 'ac'), (30805, 512, 1024, 0, 'ac'),
                                               (4097, 256, 128, 1, ''), (9001, 512, 256, 1, ''), (70001, 512, 512, 1, ''), (4100, 768, 96, 0, 'c')])
def test_wide_split_gemm_has_the_error_of_the_fp32_pipe(M, N, K, mode, gather, monkeypatch):
    """sbr_gemm_split_wide_f32 (N = 256 i, K = 32 j: the hidden layers and projectors of the C = 256 / 512 configurations on the bf16
    matrix pipe — three-way exact operand splits, six MFMA terms, fp32 accumulate; csrc/gemm_split_wide_f32.hip) through
    ops.linear_nt (mode 0, bias + ReLU, row gather and row scatter fused) and ops.matmul_nn (mode 1) against an fp64 product: error
    of the size of the fp32-pipe kernel's own, on operands spanning six decades; ragged last row group, several tiles per workgroup."""
    ops = S().ops
    monkeypatch.setattr(ops, '_WIDE_HEURISTIC', False)
    g = torch.Generator().manual_seed(M + N + K)
    rows = M if 'a' not in gather else 20011
    scale = torch.pow(10., torch.randint(-3, 3, (rows, K), generator=g).float())
    x = (torch.randn(rows, K, generator=g) * scale).to(DEV)
    a_idx = torch.randint(0, rows, (M,), generator=g, dtype=torch.int32).to(DEV) if 'a' in gather else None
    c_idx = torch.randperm(M + 7, generator=g)[:M].to(torch.int32).to(DEV) if 'c' in gather else None
    res = {}
    if mode == 0:
        w, b = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
        for flag in (True, False):
            monkeypatch.setattr(ops, '_SPLIT', flag)
            out = torch.full((M + 7 if c_idx is not None else M, N), 7.0, device=DEV)
            _lib.CALL_LOG = []
            ops.linear_nt(x, w, b, 1, a_idx=a_idx, out=out, c_idx=c_idx, n_rows=M)
            names, _lib.CALL_LOG = [n for n, _ in _lib.CALL_LOG], None
            assert ('sbr_gemm_split_wide_f32' in names) == flag, names
            res[flag] = out
        xa = x.double().cpu()[a_idx.cpu().long()] if a_idx is not None else x.double().cpu()
        pre = xa @ w.double().cpu().t() + b.double().cpu()
        ref = torch.relu(pre)
        mag = xa.abs() @ w.double().cpu().abs().t() + b.double().cpu().abs()
        if c_idx is not None:
            untouched = torch.ones(M + 7, dtype=torch.bool)
            untouched[c_idx.cpu().long()] = False
            for flag in (True, False):
                assert bool((res[flag].cpu()[untouched] == 7.0).all())          # rows no slot names are not written
                res[flag] = res[flag][c_idx.long()]
    else:
        w = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
        for flag in (True, False):
            monkeypatch.setattr(ops, '_SPLIT', flag)
            _lib.CALL_LOG = []
            res[flag] = ops.matmul_nn(x, w)
            names, _lib.CALL_LOG = [n for n, _ in _lib.CALL_LOG], None
            assert ('sbr_gemm_split_wide_f32' in names) == flag, names
        ref = x.double().cpu() @ w.double().cpu()
        mag = x.double().cpu().abs() @ w.double().cpu().abs()
    e_split = ((res[True].double().cpu() - ref).abs() / mag).max().item()
    e_f32 = ((res[False].double().cpu() - ref).abs() / mag).max().item()
    assert e_split <= max(2.0 * e_f32, 2.0 ** -22), (e_split, e_f32)
    rms_split = ((res[True].double().cpu() - ref) / mag).pow(2).mean().sqrt().item()
    rms_f32 = ((res[False].double().cpu() - ref) / mag).pow(2).mean().sqrt().item()
    assert rms_split <= max(2.0 * rms_f32, 2.0 ** -24), (rms_split, rms_f32)


@pytest.mark.gpu
@pytest.mark.parametrize('where', ['activation', 'weight'])
def test_split_gemm_on_non_finite_operands(where, monkeypatch):
    """Documented deviation of the bf16-split kernels (csrc/gemm_split_f32.hip header): an infinite operand gives NaN where the fp32
    pipe (and torch) give +-inf — the split x = x0 + x1 + x2 computes inf - inf. What is pinned: a non-finite operand poisons EXACTLY
    the outputs it poisons on the fp32 pipe (the row of an infinite activation, the column of an infinite weight: every such output
    is non-finite in both), and every other output keeps the bits of the clean product — a step that has overflowed is non-finite
    in the loss either way (train/trainer.py:215-219 reports it), nothing finite is silently changed."""
    ops = S().ops
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    M = 4100
    x, w, b = _rand(M, 128, seed=5).to(DEV), (_rand(128, 128, seed=6) / 8).to(DEV), _rand(128, seed=7).to(DEV)
    xk, wk = _rand(M, 768, seed=8).to(DEV), (_rand(128, 768, seed=9) / 16).to(DEV)
    def run(x_, w_, xk_, wk_):
        out = {}
        for flag in (True, False):
            monkeypatch.setattr(ops, '_SPLIT', flag)
            out[flag] = [ops.linear_nt(x_, w_, b, 1), ops.matmul_nn(x_, w_), ops.linear_nt(xk_, wk_, b, 0)]
        return out
    clean = run(x, w, xk, wk)
    x2, w2, xk2, wk2 = x.clone(), w.clone(), xk.clone(), wk.clone()
    if where == 'activation':
        x2[17, 3], x2[4099, 127], xk2[17, 700] = float('inf'), float('-inf'), float('inf')
        bad = lambda t, which: (torch.zeros_like(t, dtype=torch.bool).index_fill_(0, torch.tensor([17, 4099] if which < 2 else [17], device=DEV), True))
    else:
        w2[5, 9], wk2[5, 9] = float('inf'), float('inf')
        # NT: W[n, k] -> output column n = 5; NN: W[k, n] -> output column n = 9
        bad = lambda t, which: (torch.zeros_like(t, dtype=torch.bool).index_fill_(1, torch.tensor([9 if which == 1 else 5], device=DEV), True))
    dirty = run(x2, w2, xk2, wk2)
    for which in range(3):
        m = bad(clean[True][which], which)
        for flag in (True, False):
            d, c = dirty[flag][which], clean[flag][which]
            if which == 0 and not flag:                               # relu epilogue on the fp32 pipe: relu(-inf) = 0, as in torch
                assert bool((~torch.isfinite(d[m]) | (d[m] == 0)).all()) and not bool(torch.isfinite(d[m]).all())
            else:                                                     # (relu keeps a NaN a NaN: sbr_relu, torch.relu)
                assert not bool(torch.isfinite(d[m]).any()), (which, flag)
            assert torch.equal(d[~m], c[~m]), (which, flag)           # untouched outputs: the bits of the clean product


@pytest.mark.gpu
@pytest.mark.parametrize('M,K', [(4096, 768), (5000, 256), (9001, 1024), (45801, 256), (36000, 256), (52000, 256), (70001, 256)])
def test_split_projector_gemm_has_the_error_of_the_fp32_pipe(M, K, monkeypatch):
    """sbr_gemm_split_proj_f32 (the dense modality projector on the bf16 matrix pipe: K walked in chunks of 128, row gather and row
    scatter fused) against an fp64 product, next to the fp32-pipe kernel on the same call: gathered rows with repeats, scattered
    output rows, bias + the three activation kinds, a ragged last block; operands span six decades. The long shapes cover the
    work-item map on 256 workgroups: a last round whose blocks beyond one per SIMD are split by columns between two waves (1,432
    and 1,125 blocks), one that fills both waves of the SIMDs (1,625), and two rounds (2,188)."""
    ops = S().ops
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    g = torch.Generator().manual_seed(5)
    n_src = 3000
    scale = torch.pow(10., torch.randint(-3, 3, (n_src, K), generator=g).float())
    x = (_rand(n_src, K, seed=51) * scale).to(DEV)
    w, b = (_rand(128, K, seed=52) / 16).to(DEV), _rand(128, seed=53).to(DEV)
    a_idx = torch.randint(0, n_src, (M,), generator=g, dtype=torch.int32).to(DEV)
    c_idx = torch.randperm(M + 7, generator=g)[:M].to(torch.int32).to(DEV)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, '_SPLIT', flag)
        outs = []
        for act in (0, 1, 2):
            out = torch.full((M + 7, 128), 123.0, device=DEV)
            ops.linear_nt(x, w, b, act, a_idx=a_idx, out=out, c_idx=c_idx, n_rows=M)
            outs.append(out)
        outs.append(ops.linear_nt(x, w, None, 0, a_idx=a_idx))            # no bias, no scatter
        res[flag] = outs
    xd, wd, bd = x.double().cpu()[a_idx.cpu().long()], w.double().cpu(), b.double().cpu()
    pre = xd @ wd.t() + bd
    absum = xd.abs() @ wd.abs().t() + bd.abs()
    ci = c_idx.cpu().long()
    untouched = torch.ones(M + 7, dtype=torch.bool); untouched[ci] = False
    for i, ref in ((0, pre), (1, torch.relu(pre)), (3, xd @ wd.t())):
        got = {f: (res[f][i].double().cpu()[ci] if i < 3 else res[f][i].double().cpu()) for f in (True, False)}
        mag = absum if i < 3 else xd.abs() @ wd.abs().t()
        e_split = ((got[True] - ref).abs() / mag).max().item()
        e_f32 = ((got[False] - ref).abs() / mag).max().item()
        assert e_split <= max(2.0 * e_f32, 2.0 ** -21), (i, e_split, e_f32)
        if i < 3:
            assert bool((res[True][i].cpu()[untouched] == 123.0).all())        # rows outside c_idx are not written
    assert bool(((res[True][2].double().cpu()[ci] - res[False][2].double().cpu()[ci]).abs() <= 2.0 ** -19 * absum + 1e-6).all())


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,K,gather', [(128, 128, 4096, ''), (128, 128, 9001, ''), (128, 128, 90112, 'a'), (128, 768, 8200, 'b'),
                                          (128, 256, 20011, 'ab'), (128, 768, 45824, 'b'), (128, 128, 600000, ''), (256, 128, 9001, 'b'),
                                          (512, 512, 30011, 'a')])
def test_split_tn_kernel_has_the_error_of_the_fp32_pipe(M, N, K, gather, monkeypatch):
    """sbr_gemm_tn_f32 on the weight-gradient shapes (M and N multiples of 128, K >= 4096 rows) runs on the bf16 matrix pipe
    (csrc/gemm_split_tn_f32.hip: exact three-way split of both operands, six MFMA terms): against an fp64 product, next to the
    fp32-pipe ring kernel (SBR_TN_SPLIT=0) on the same call — plain and row-gathered operands (either side), operands spanning six
    decades, K ranges that end inside a 32-row chunk, more than one round of workgroups (K = 600,000; 512 x 512), the deferred-slab path."""
    ops = S().ops
    g = torch.Generator().manual_seed(19)
    n_src = 5000
    scale_a = torch.logspace(-3, 3, M).unsqueeze(0)
    dz = (_rand(n_src if 'a' in gather else K, M, seed=71) * scale_a).to(DEV)
    x = (_rand(n_src if 'b' in gather else K, N, seed=72) * torch.logspace(-2, 2, N).unsqueeze(0)).to(DEV)
    a_idx = torch.randint(0, n_src, (K,), generator=g, dtype=torch.int32).to(DEV) if 'a' in gather else None
    b_idx = torch.randint(0, n_src, (K,), generator=g, dtype=torch.int32).to(DEV) if 'b' in gather else None
    ad = dz.double().cpu()[a_idx.cpu().long()] if 'a' in gather else dz.double().cpu()
    xd = x.double().cpu()[b_idx.cpu().long()] if 'b' in gather else x.double().cpu()
    want = ad.t() @ xd
    mag = ad.abs().t() @ xd.abs()
    got = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('SBR_TN_SPLIT', flag)
        got[flag] = ops.matmul_tn(dz, x, a_idx=a_idx, b_idx=b_idx, n_rows=K).double().cpu()
    e_split = ((got['1'] - want).abs() / mag).max().item()
    e_ring = ((got['0'] - want).abs() / mag).max().item()
    assert not torch.equal(got['1'], got['0'])                 # the split kernel did run
    assert e_split <= max(1.5 * e_ring, 2.0 ** -21), (e_split, e_ring)
    # twice the same bits (fixed summation order), and the deferred slabs + shared reducer give them too
    monkeypatch.setenv('SBR_TN_SPLIT', '1')
    assert torch.equal(ops.matmul_tn(dz, x, a_idx=a_idx, b_idx=b_idx, n_rows=K).double().cpu(), got['1'])
    d = ops.DeferredTN()
    out = torch.empty(M, N, device=DEV)
    d.matmul_tn('t', dz, x, a_idx=a_idx, b_idx=b_idx, n_rows=K, out=out)
    d.finish()
    assert torch.equal(out.double().cpu(), got['1'])


@pytest.mark.gpu
@pytest.mark.parametrize('n_fin', [1, 3, 8])
def test_slab_reducer_also_finishes_pending_column_sums(n_fin):
    """DeferredTN.finish(colred) = sbr_splitk_reduce_multi_fin: one launch sums the split-K slabs of the pending dW products AND turns
    pending column-reduction workspaces into their float vectors — the same bits as the two launches (finish() + colred_finish), the
    replicas left zeroed; widths larger than the largest product's slice count included."""
    ops = S().ops
    dz, x = _rand(9000, 128, seed=81).to(DEV), _rand(9000, 256, seed=82).to(DEV)
    widths = [128, 512, 64, 1024, 128, 256, 128, 4][:n_fin]
    dys = [_rand(5000, C, seed=90 + i).to(DEV) for i, C in enumerate(widths)]
    ys = [_rand(5000, C, seed=70 + i).to(DEV) for i, C in enumerate(widths)]
    res = {}
    for fused in (False, True):
        pend = []
        for dy, y in zip(dys, ys):
            ws = ops.new_colsum_ws(DEV, dy.shape[1])
            ops.act_grad_colsum(dy, y, 1, ws)
            pend.append((ws, torch.full((dy.shape[1],), 7.0, device=DEV)))
        d = ops.DeferredTN()
        out = torch.empty(128, 256, device=DEV)
        d.matmul_tn('t', dz, x, out=out)
        if fused:
            assert d.finish(pend) is True
        else:
            assert not d.finish()
            ops.colred_finish(pend)
        res[fused] = (out.clone(), [o.clone() for _, o in pend])
        for ws, o in pend:
            assert bool((ws[o.numel():] == 0).all())            # replicas left zeroed
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('D', [128, 256])
def test_fused_f16_scorer_long_tailed_exclusion_rows(D):
    """The exclusion event stream of the narrow-wave kernel (csrc/score_topk_f16_n.hip) on rows of very different lengths: users
    without exclusions, users with thousands (many events per item tile, several event quads per tile), users whose WHOLE shard is
    excluded (fewer than k candidates: -inf / -1 padding), a shuffled user -> CSR-row map, and a shard window in the middle of the
    catalogue — against the fp64 product of the same fp16 values with the mask applied densely."""
    ops = S().ops
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    Bu, I_all, off, I, k = 2500, 9000, 3000, 4000, 20
    u = (_rand(Bu, D, seed=90) / 4).half()
    it = (_rand(I_all, D, seed=91) / 4).half()
    deg = np.minimum(I_all, rng.lognormal(3.0, 1.6, size=Bu).astype(np.int64))
    deg[:40] = 0                                               # no exclusions at all
    deg[40:48] = I_all                                         # everything excluded
    deg[48:60] = rng.integers(3000, 6000, size=12)             # thousands
    rows = np.repeat(np.arange(Bu), deg)
    cols = np.concatenate([np.sort(rng.choice(I_all, size=d, replace=False)) for d in deg]) if rows.size else np.zeros(0, np.int64)
    m = sp.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(Bu, I_all))
    m.sort_indices()
    perm = rng.permutation(Bu)                                 # scored row b uses CSR row perm[b]
    indptr = torch.from_numpy(m.indptr.astype(np.int64)).to(DEV)
    indices = torch.from_numpy(m.indices.astype(np.int32)).to(DEV)
    uidx = torch.from_numpy(perm.astype(np.int64)).to(DEV)
    val, idx = ops.score_topk_f16(u.to(DEV), it[off:off + I].contiguous().to(DEV), k, uidx, indptr, indices, item_offset=off)
    ref = u.double() @ it[off:off + I].double().t()
    dense = torch.from_numpy(m[perm][:, off:off + I].toarray() != 0)
    ref[dense] = -float('inf')
    tv, ti = torch.topk(ref, k, sorted=True)
    val, idx = val.cpu(), idx.cpu()
    finite = ~torch.isinf(tv)
    close(val[finite], tv[finite], rtol=1e-5, atol=1e-5, what='fused values')
    assert bool(torch.isinf(val[~finite]).all()) and bool((val[~finite] < 0).all())
    assert bool((idx[~finite] == -1).all()), 'slots behind the last candidate must carry index -1'
    got = torch.gather(ref, 1, (idx.long() - off).clamp_min(0))
    close(got[finite], tv[finite], rtol=1e-5, atol=1e-5, what='scores of the selected items')
    assert bool(((idx[finite] >= off) & (idx[finite] < off + I)).all())


@pytest.mark.gpu
@pytest.mark.parametrize('M,act', [(4096, 0), (9001, 1)])
def test_split_gemm_statistics_epilogue_feeds_batchnorm(M, act, monkeypatch):
    """sbr_gemm_split_f32 (mode 0) with a statistics workspace: the per-column sums / sums of squares of its output, completed by
    sbr_bn_finalize_stats, are the batch mean / rstd and running-statistics update that sbr_bn_train_stats computes from a pass of
    its own over the same output (ragged last block, with and without activation)."""
    mod = S()
    from importlib import import_module
    ops, lib_ = mod.ops, import_module(mod.ops.__name__.rsplit('.', 1)[0] + '._lib')
    monkeypatch.setattr(ops, '_SPLIT_MIN_ROWS', 1)
    x, w, b = _rand(M, 128, seed=71).to(DEV), (_rand(128, 128, seed=72) / 8).to(DEV), _rand(128, seed=73).to(DEV)
    out = torch.empty(M, 128, device=DEV)
    assert ops.linear_nt_stats_ok(x, w, out)
    ws = torch.zeros(ops.COLRED_WS_FACTOR * 2 * 128, device=DEV, dtype=torch.float64)
    ops.linear_nt(x, w, b, act, out=out, stats_ws=ws)
    plain = ops.linear_nt(x, w, b, act)
    assert torch.equal(out, plain)                              # the epilogue does not change what is stored
    res = {}
    for tag in ('folded', 'own pass'):
        rm, rv, nbt = torch.zeros(128, device=DEV), torch.ones(128, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        mean, rstd = torch.empty(128, device=DEV), torch.empty(128, device=DEV)
        if tag == 'folded':
            lib_.call('sbr_bn_finalize_stats', M, 128, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                      ws.data_ptr(), ops.BN_EPS, ops.BN_MOMENTUM, lib_.stream())
        else:
            ws2 = torch.zeros_like(ws)
            lib_.call('sbr_bn_train_stats', out.data_ptr(), M, 128, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), mean.data_ptr(),
                      rstd.data_ptr(), ws2.data_ptr(), ops.BN_EPS, ops.BN_MOMENTUM, lib_.stream())
        res[tag] = [t.cpu() for t in (mean, rstd, rm, rv, nbt)]
    od = out.double().cpu()
    close(res['folded'][0], od.mean(0).float(), rtol=1e-5, atol=1e-6, what='batch mean')
    close(res['folded'][1], (1.0 / torch.sqrt(od.var(0, unbiased=False) + ops.BN_EPS)).float(), rtol=1e-5, atol=1e-6, what='batch rstd')
    for a_, b_, what in zip(res['folded'], res['own pass'], ('mean', 'rstd', 'running mean', 'running var', 'batches')):
        close(a_.double(), b_.double(), rtol=1e-6, atol=1e-7, what=what)
    assert float(ws[2 * 128:].abs().max()) == 0.0                # replicas left zeroed for the next use
    # the finalisation by the GEMM's last workgroup (bn_fin): one launch, the bits of the two above, workspace and counter left zeroed,
    # a second call on the same workspace / counter the same again
    arrive = torch.zeros(1, dtype=torch.int64, device=DEV)
    for rep in range(2):
        rm, rv, nbt = torch.zeros(128, device=DEV), torch.ones(128, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        mean, rstd = torch.empty(128, device=DEV), torch.empty(128, device=DEV)
        out2 = torch.empty(M, 128, device=DEV)
        ops.linear_nt(x, w, b, act, out=out2, stats_ws=ws, bn_fin=(arrive, rm, rv, nbt, mean, rstd, ops.BN_EPS, ops.BN_MOMENTUM))
        assert torch.equal(out2, plain)
        for a_, b_, what in zip((mean, rstd, rm, rv, nbt), res['folded'], ('mean', 'rstd', 'running mean', 'running var', 'batches')):
            assert torch.equal(a_.cpu(), b_), what
        assert float(ws[2 * 128:].abs().max()) == 0.0 and int(arrive.item()) == 0
