"""GPU: the two-pass fused scorer (csrc/score_topk_f16_2p.hip: group-maxima pass, pair selection, grouped re-scoring, final selection)
against the one-pass kernel (csrc/score_topk_f16_n.hip, itself pinned against the fp32 GEMM + exact top-k route and through it against
eval/eval.py:216-222) — the two routes must return the SAME lists bit for bit (same MFMA chain per score, same ordering rule: score
desc, item index asc) — and directly against the fp32 route on samples."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def S():
    import sibrar_amd
    return sibrar_amd


def _excl(U, I_total, per, seed, heavy=()):
    rng = np.random.default_rng(seed)
    rows, cols = [np.repeat(np.arange(U), per)], [rng.integers(0, I_total, size=U * per)]
    for (u, n) in heavy:                                       # users with very long exclusion rows
        rows.append(np.full(n, u))
        cols.append(rng.choice(I_total, size=n, replace=False))
    m = sp.csr_matrix((np.ones(sum(len(r) for r in rows), dtype=np.int8), (np.concatenate(rows), np.concatenate(cols))), shape=(U, I_total))
    m.sum_duplicates()
    m.sort_indices()
    return S().evaluation._csr_to_device(m, DEV)


def _both(u16, i16, k, users=None, ex=None, off=0):
    ops = S().ops
    out = []
    for route in (1, 2):
        prev = ops.score_topk_route(route)
        try:
            if ex is None:
                out.append(ops.score_topk_f16(u16, i16, k, item_offset=off))
            else:
                out.append(ops.score_topk_f16(u16, i16, k, users, ex[0], ex[1], item_offset=off))
        finally:
            ops.score_topk_route(prev)
    torch.cuda.synchronize()
    return out


def _same(a, b, what=''):
    (v1, i1), (v2, i2) = a, b
    bad = (i1 != i2).any(dim=1) | (v1 != v2).any(dim=1)
    assert not bool(bad.any()), f'{what}: {int(bad.sum())} users differ, first {int(bad.nonzero()[0])}: one-pass {i1[bad][0].tolist()} two-pass {i2[bad][0].tolist()}'


def _near(got, ref, sc, off=0):
    """Against the fp32 GEMM route: the same lists, except that two items whose fp32 scores differ by rounding only (the GEMM sums in
    another order than the MFMA chain) may trade places — every listed item's fp32 score is within 2e-6 of the reference list's score at
    that rank, and no excluded item is listed."""
    (gv, gi), (rv, ri) = got, ref
    n = rv.shape[0]
    picked = torch.gather(sc[:n], 1, (gi[:n].long() - off).clamp_min(0))
    assert bool((picked > -float('inf')).all()), 'an excluded item was listed'
    assert bool(((picked - rv).abs() <= 2e-6 * rv.abs().clamp_min(1.0)).all()), 'a listed item is not a top-k item of the fp32 route'
    assert float((gi[:n].long() != ri.long() + off).float().mean()) < 0.01, 'more than 1 % of the positions differ from the fp32 route'


@pytest.mark.parametrize('U,I,D,k,per,off', [(3000, 20000, 128, 20, 30, 0), (2500, 16384, 64, 10, 0, 0), (1100, 9000, 256, 20, 25, 5000),
                                             (40000, 30011, 128, 20, 50, 0), (9000, 12345, 256, 32, 10, 777), (777, 8192, 128, 1, 5, 0),
                                             (33000, 8700, 64, 20, 40, 100)])
def test_two_pass_scorer_equals_the_one_pass_kernel(U, I, D, k, per, off):
    """Random representations, exclusions (per user `per` random items of the whole catalogue; the shard starts at `off`), catalogue
    sizes that end inside a supertile / a tile, user counts with remainder units and part waves, every supported D, k = 1 .. 32."""
    g = torch.Generator().manual_seed(U + I)
    u16 = (torch.randn(U, D, generator=g) / 8).half().to(DEV)
    i16 = (torch.randn(I, D, generator=g) / 8).half().to(DEV)
    users = torch.arange(U, device=DEV)
    ex = _excl(U, off + I + 100, per, U, heavy=((5, 3000), (U - 1, 6000))) if per else None
    one, two = _both(u16, i16, k, users, ex, off)
    _same(one, two, f'{U}x{I}x{D}')
    # and a sample directly against the fp32 GEMM -> mask -> exact top-k route
    n = min(U, 512)
    sc = u16[:n].float() @ i16.float().t()
    if ex is not None:
        S().ops.mask_scores_(sc, users[:n], ex[0], ex[1], item_offset=off)
    rv, ri = S().ops.topk_rows(sc, k)
    _near((two[0][:n], two[1][:n]), (rv, ri), sc, off)


def test_two_pass_scorer_with_massive_ties_and_degenerate_users():
    """Hard users take the exact slow path of the final kernel: all-equal scores (every group ties at the bound), a user whose row is
    zero, users with fewer than k scoreable items (everything else excluded), duplicated items (exact ties between groups)."""
    g = torch.Generator().manual_seed(3)
    U, I, D, k = 600, 10000, 128, 20
    u16 = (torch.randn(U, D, generator=g) / 8).half()
    i16 = (torch.randn(I, D, generator=g) / 8).half()
    u16[7] = 0                                                   # every score 0: the first k items win
    i16[2000:6000] = i16[0:4000].clone()                                 # 4,000 exact duplicates: ties across groups and supertiles
    u16, i16 = u16.to(DEV), i16.to(DEV)
    rng = np.random.default_rng(1)
    rows, cols = [], []
    for u in range(U):
        if u == 11:                                              # 5 scoreable items only
            c = np.setdiff1d(np.arange(I), [3, 4000, 4001, 9998, 9999])
        elif u == 12:                                            # nothing scoreable
            c = np.arange(I)
        else:
            c = rng.integers(0, I, size=20)
        rows.append(np.full(len(c), u)); cols.append(c)
    m = sp.csr_matrix((np.ones(sum(len(r) for r in rows), dtype=np.int8), (np.concatenate(rows), np.concatenate(cols))), shape=(U, I))
    m.sum_duplicates(); m.sort_indices()
    ex = S().evaluation._csr_to_device(m, DEV)
    users = torch.arange(U, device=DEV)
    one, two = _both(u16, i16, k, users, ex)
    _same(one, two, 'ties')
    assert two[1][12].tolist() == [-1] * k and two[1][11, 5:].tolist() == [-1] * (k - 5)
    assert sorted(two[1][11, :5].tolist()) == [3, 4000, 4001, 9998, 9999]
    # constant catalogue: every user is hard
    i_const = i16[:1].expand(9000, D).contiguous()
    one, two = _both(u16[:100], i_const, k)
    _same(one, two, 'constant catalogue')
    assert two[1][0].tolist() == list(range(k))


def test_two_pass_scorer_at_the_bench_shapes():
    """BASELINE configs[1] / [4] scoring shapes (100k x 50k x 128 with ~50 exclusions per user; 100k x 25k x 256, shard offset 75,000)."""
    for (U, I, D, off) in ((100_000, 50_000, 128, 0), (100_000, 25_000, 256, 75_000)):
        g = torch.Generator().manual_seed(D)
        u16 = (torch.randn(U, D, generator=g) / 16).half().to(DEV)
        i16 = (torch.randn(I, D, generator=g) / 16).half().to(DEV)
        users = torch.arange(U, device=DEV)
        ex = _excl(U, off + I, 50, D, heavy=((17, 3345),))
        one, two = _both(u16, i16, 20, users, ex, off)
        _same(one, two, f'{U}x{I}x{D}')


def test_both_scorer_routes_with_a_user_index_map():
    """``u_idx`` maps the scored rows to rows of the exclusion CSR (an evaluation chunk of a split whose users are not 0 .. U - 1): both
    routes read the exclusions of the mapped user, and agree with the fp32 route."""
    g = torch.Generator().manual_seed(17)
    U, I, D, k = 1500, 9000, 128, 10
    n_all = 5000
    u16 = (torch.randn(U, D, generator=g) / 8).half().to(DEV)
    i16 = (torch.randn(I, D, generator=g) / 8).half().to(DEV)
    ex = _excl(n_all, I, 40, 17, heavy=((4321, 2500),))
    users = torch.randperm(n_all, generator=g)[:U].to(DEV)          # int64 user ids in scoring order
    users[7] = 4321
    one, two = _both(u16, i16, k, users, ex)
    _same(one, two, 'u_idx')
    sc = u16.float() @ i16.float().t()
    S().ops.mask_scores_(sc, users, ex[0], ex[1])
    rv, ri = S().ops.topk_rows(sc, k)
    _near(two, (rv, ri), sc)
