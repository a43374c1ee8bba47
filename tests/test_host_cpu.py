"""CPU (-m "not gpu"): host logic of the product — C-ABI library loads and exports every declared symbol (no compute
calls), bit-exact samplers against the oracle and the golden streams, config parsing, state_dict contract, flat-buffer
optimizer plumbing, and the multi-GPU paths on 2 gloo ranks."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from golden_util import MANIFEST, load, world, host_dataset, state_dict, side_cfg, U, I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from importlib import import_module
    _lib = import_module('sibrar---single-branch-recommender_amd._lib')
    protos = _lib.parse_header()
    assert len(protos) >= 36
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r' T (sbr_\w+)', out))
    assert set(protos) <= exported, set(protos) - exported
    assert exported <= set(protos), f'exported but undeclared: {exported - set(protos)}'
    assert handle.sbr_abi_version() == 4
    assert handle.sbr_last_error() is not None


def test_product_library_carries_no_lab_switches():
    """The timing-only ablations of the fused scorer (wrong results by design) and the environment switches that selected them are
    compiled only into lab builds (-DSBR_LAB, tools/lab/build_scorer_variants.sh): the product library holds ONE instantiation of the
    scorer kernel per supported D and never mentions the variables."""
    from importlib import import_module
    _lib = import_module('sibrar---single-branch-recommender_amd._lib')
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run(['nm', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    kernels = set(re.findall(r' [VW] (_Z\d+score_topk_f16_n_kernelI\w+)', out))
    assert len(kernels) == 3, kernels                                   # D = 64, 128, 256
    assert all(re.search(r'ELi0ELb[01]EEv', k) for k in kernels), kernels     # template argument DBG = 0
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'SBR_ST_DEBUG' not in blob and b'SBR_ST_PRE' not in blob


def test_ops_fail_loudly_without_gpu_tensors():
    import sibrar_amd as S
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        S.ops.LinearActFn.apply(torch.randn(4, 3), torch.randn(2, 3), None, 0)
    ds = host_dataset(world(load('g4_full_net')))
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict({
        'shared_common_dim': 8, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
        'item': {'features': [{'feature_name': 'text'}], 'single_branch_hidden_layers': [], 'preference_hidden_layers': [],
                 'common_modality_dim': 8}}), ds)
    with pytest.raises(RuntimeError, match='CUDA'):
        net(torch.zeros(2, dtype=torch.long), torch.zeros(2, 3, dtype=torch.long))


def test_missing_library_raises(monkeypatch):
    from importlib import import_module
    _lib = import_module('sibrar---single-branch-recommender_amd._lib')
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsibrar_hip.so')
    with pytest.raises(_lib.SibrarHipError, match='no CPU fallback'):
        _lib.lib()


@pytest.mark.parametrize('case', MANIFEST['g7_row_wise_sample']['cases'], ids=lambda c: c['name'])
def test_vectorised_modality_sampler_matches_reference_stream(case):
    import sibrar_amd as S
    z = load('g7_row_wise_sample')
    n = case['name']
    a = case['a']
    if case['central'] is not None:
        a = [case['central']] + case['central_others_order']
    reg = 'central_modality' if case['central'] else ('no_regularization' if case['k'] == 1 else 'pairwise_single')
    rng = np.random.default_rng(42)
    rows = int(np.prod(case['size']))
    for call in range(2):
        pos = S.sampling.sample_modalities(rng, a, rows, reg, case['central'])
        got = np.array(a)[pos].reshape(tuple(case['size']) + (case['k'],))
        assert (got == z[f'{n}/call{call}']).all()


def test_vectorised_sampler_long_streams_vs_oracle():
    import sibrar_amd as S
    from oracle import sampling_ref
    for n_mod in (2, 3, 5):
        for k in (1, 2):
            a = [f'm{i}' for i in range(n_mod)]
            r1, r2 = np.random.default_rng(7), np.random.default_rng(7)
            ref = sampling_ref.row_wise_sample(a, (3000,), k=k, rng=r1)
            got = S.sampling.sample_modality_ids(r2, 3000, n_mod, k)
            assert (np.array(a)[got] == ref).all()
            assert r1.integers(0, 1 << 30) == r2.integers(0, 1 << 30)      # generators left in the same state


def test_negative_sampling_loaders_match_reference_streams():
    import sibrar_amd as S
    from oracle import sampling_ref
    z = load('g6_neg_sampling')
    w = world(z)
    meta = MANIFEST['g6_neg_sampling']
    ds = S.SyntheticDataset.__new__(S.SyntheticDataset)
    ds.interaction_matrix = w['inter'].tocoo()
    ds.user_sampling_matrix = w['inter']
    ds.items_in_split = np.arange(I)
    ds.n_items = I
    ds.n_negative_samples = meta['n_neg']
    ds.negative_sampling_strategy = 'uniform_recbole'
    assert (ds.interaction_matrix.row == z['coo_row']).all() and (ds.interaction_matrix.col == z['coo_col']).all()
    for strategy, key in (('uniform_recbole', 'recbole'), ('uniform', 'uniform')):
        sampling_ref.reproducible(42)
        loader = S.NegativeSamplingDataLoader(ds, batch_size=meta['batch_size'], shuffle=True, strategy=strategy)
        for b, (u, i, l) in enumerate(loader):
            if b >= 3:
                break
            assert u.dtype == torch.int64 and i.dtype == torch.int64 and l.dtype == torch.float64
            assert (u.numpy() == z[f'{key}/u{b}']).all() and (i.numpy() == z[f'{key}/i{b}']).all()
            assert (l.numpy() == z[f'{key}/l{b}']).all()


def test_config_parsing_and_state_dict_contract():
    import sibrar_amd as S
    z = load('g4_full_net')
    for case in MANIFEST['g4_full_net']['cases'][::5]:
        cfg = S.SingleBranchNetConfig.from_dict({'shared_common_dim': case['shared_common_dim'], 'user': side_cfg(case['user']),
                                                 'item': side_cfg(case['item'])})
        assert cfg.is_item_sb_module and cfg.is_user_sb_module == (case['user_kind'] == 'entity')
        net = S.SingleBranchNet(cfg, host_dataset(world(z)))
        ref = state_dict(z, f"{case['name']}/sd0/")
        mine = net.state_dict()
        assert list(mine.keys()) == list(ref.keys())
        for k in ref:
            assert tuple(mine[k].shape) == tuple(ref[k].shape) and mine[k].dtype == ref[k].dtype, k
        net.load_state_dict(ref, strict=True)
        # the CSR projector weight keeps its [C, n_cols] shape but is stored column-major
        w = net.item_embedding_module.modality_modules['interactions'].pre_embedding_layers.layers.linear_0.weight
        assert w.t().is_contiguous() and torch.equal(w.detach(), ref['item_embedding_module.modality_modules.interactions.'
                                                                     'pre_embedding_layers.layers.linear_0.weight'])
    with pytest.raises(ValueError, match='at least one feature'):
        S.SingleBranchNetEntity('item', {}, S.SingleBranchNetEntityConfig.from_dict(
            {'features': [], 'single_branch_hidden_layers': [], 'preference_hidden_layers': [], 'common_modality_dim': 4}), 4)
    with pytest.raises(ValueError, match='not supported'):
        ds = host_dataset(world(z))
        S.SingleBranchNetEntity('item', ds.item_features, S.SingleBranchNetEntityConfig.from_dict(
            {'features': [{'feature_name': 'text'}], 'single_branch_hidden_layers': [], 'preference_hidden_layers': [],
             'common_modality_dim': 4, 'aggregation_fn': 'median'}), 4)


def test_init_rules_follow_reference():
    """train/utils.py:5-13: kaiming-uniform(relu) bound sqrt(6/fan_in), zero bias, Embedding std 0.1/dim, EmbeddingBag default."""
    import sibrar_amd as S
    torch.manual_seed(0)
    ds = host_dataset(world(load('g4_full_net')))
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict({
        'shared_common_dim': 8, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
        'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'genres'}, {'feature_name': 'interactions'}],
                 'single_branch_hidden_layers': [8], 'preference_hidden_layers': [], 'common_modality_dim': 8}}), ds)
    sd = net.state_dict()
    w = sd['item_embedding_module.modality_modules.text.pre_embedding_layers.layers.linear_0.weight']
    assert w.abs().max() <= (6 / 16) ** 0.5 + 1e-6 and w.abs().max() > 0.5 * (6 / 16) ** 0.5
    assert torch.count_nonzero(sd['item_embedding_module.modality_modules.text.pre_embedding_layers.layers.linear_0.bias']) == 0
    assert sd['user_embedding_module.embedding_layer.weight'].std() < 0.1 / 8 * 1.5
    bag = sd['item_embedding_module.modality_modules.genres.embedding_layer.weight']
    assert torch.count_nonzero(bag[-1]) == 0 and 0.5 < bag[:-1].std() < 1.5


def test_merge_topk_is_exact():
    import sibrar_amd as S
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(50, 400, generator=g)
    scores[:, 100:110] = scores[:, 300:310]            # ties across shards
    k, world_size = 20, 4
    vals, idxs = [], []
    for r in range(world_size):
        lo, hi = S.parallel.item_shard(400, r, world_size)
        v, i = torch.topk(scores[:, lo:hi], k, sorted=True)
        vals.append(v)
        idxs.append((i + lo).int())
    v, i = S.parallel.merge_topk(torch.cat(vals, 1), torch.cat(idxs, 1), k)
    tv, _ = torch.topk(scores, k, sorted=True)
    assert torch.equal(v, tv)
    assert torch.equal(torch.gather(scores, 1, i.long()), tv)
    same = v[:, 1:] == v[:, :-1]
    assert (i[:, 1:][same] > i[:, :-1][same]).all()


_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import sibrar_amd as S
dist.init_process_group('gloo', init_method='file://' + os.environ['SBR_RDZV_FILE'], rank=int(os.environ['RANK']),
                        world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
# (1) flat gradient all-reduce == mean of the per-rank gradients
g = torch.arange(10, dtype=torch.float32) * (rank + 1)
S.parallel.all_reduce_flat_(g)
assert torch.allclose(g, torch.arange(10, dtype=torch.float32) * (1 + 2) / 2), g
# (2) every rank draws the SAME global batch and keeps its slice; the union is the global batch
ds = S.SyntheticDataset(120, 80, 1500, seed=1, n_negative_samples=3)
torch.manual_seed(5); np.random.seed(5)
full = next(iter(S.NegativeSamplingDataLoader(ds, batch_size=32)))
torch.manual_seed(5); np.random.seed(5)
mine = next(iter(S.NegativeSamplingDataLoader(ds, batch_size=32, rank=rank, world=world)))
assert torch.equal(mine[1], full[1][rank::world]) and torch.equal(mine[0], full[0][rank::world])
gathered = [torch.empty_like(mine[1]) for _ in range(world)]
dist.all_gather(gathered, mine[1])
glob = torch.empty_like(full[1]); glob[0::2], glob[1::2] = gathered[0], gathered[1]
assert torch.equal(glob, full[1])
# (3) item-sharded top-k: all-gather + merge == global top-k
gen = torch.Generator().manual_seed(9)
scores = torch.randn(16, 101, generator=gen)
lo, hi = S.parallel.item_shard(101, rank, world)
v, i = torch.topk(scores[:, lo:hi], 5, sorted=True)
mv, mi = S.parallel.all_gather_topk(v, (i + lo).int(), 5)
tv, ti = torch.topk(scores, 5, sorted=True)
assert torch.equal(mv, tv) and torch.equal(mi.long(), ti)
# (4) the assembly of evaluation.evaluate_recommender_algorithm for a shard SHORTER than the list (5 items over 2 ranks, top-4):
# empty slots (-inf, -1) behind a shard's items, merged lists == global top-4 with positions in items_in_split
small = torch.randn(16, 5, generator=gen)
lo, hi = S.parallel.item_shard(5, rank, world)
kl = min(4, hi - lo)
v, i = torch.topk(small[:, lo:hi], kl, sorted=True)
i = (i + lo).int()
v = torch.cat([v, torch.full((16, 4 - kl), -float('inf'))], 1)
i = torch.cat([i, torch.full((16, 4 - kl), -1, dtype=torch.int32)], 1)
mv, mi = S.parallel.all_gather_topk(v, i, 4)
tv, ti = torch.topk(small, 4, sorted=True)
assert torch.equal(mv, tv) and torch.equal(mi.long(), ti)
dist.barrier()
print('rank', rank, 'ok')
'''


def test_two_rank_gloo_data_parallel_and_item_sharding(tmp_path):
    """Rendezvous through a FileStore (no port to pick, nothing to retry: a failure of the ranks is a failure of the test)."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, SBR_RDZV_FILE=str(tmp_path / 'rdzv'), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r} failed:\n{o}'
        assert f'rank {r} ok' in o


@pytest.mark.parametrize('high,n', [(50000, 81920), (1, 4000), (2, 5000), (3299, 2816), (65536, 4096), (65537, 4097), (2 ** 32, 3000)])
def test_native_legacy_randint_matches_numpy_stream(high, n):
    """csrc/host_rng.hip (host code, no GPU): same values as np.random.randint(0, high, n) on the global legacy generator AND the
    same generator state afterwards (the draws that follow are identical)."""
    from importlib import import_module
    import sibrar_amd
    sampling = import_module(sibrar_amd.ops.__name__.rsplit('.', 1)[0] + '.sampling')
    np.random.seed(high % 1000 + n)
    np.random.randint(0, 7, size=n % 601)                    # arbitrary position inside the 624-word state
    st = np.random.get_state()
    want, after = np.random.randint(0, high, size=n), np.random.randint(0, 1000, size=9)
    np.random.set_state(st)
    got, after2 = sampling.legacy_randint(high, n), np.random.randint(0, 1000, size=9)
    assert got.dtype == want.dtype and np.array_equal(got, want) and np.array_equal(after, after2)


def _csr_eq(m, z, key):
    import scipy.sparse as sp
    m = sp.csr_matrix(m).astype(np.int64)
    m.sum_duplicates()
    m.sort_indices()
    assert tuple(z[key + '/shape']) == m.shape, key
    assert np.array_equal(m.indptr, z[key + '/indptr']) and np.array_equal(m.indices, z[key + '/indices']), key
    assert np.array_equal(np.asarray(m.data).astype(np.int64), z[key + '/data']), key


@pytest.mark.parametrize('name', ['split_random', 'split_cold_item'])
@pytest.mark.parametrize('split', ['train', 'val', 'test'])
def test_split_loader_matches_reference_datasets(name, split):
    """sibrar_amd.load_split_dataset on the committed on-disk fixtures == what the reference's TrainRecDataset /
    FullEvalDataset expose for the same directories (g12, generated with the real reference by make_golden_split.py)."""
    import json
    import sibrar_amd as S
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'g12_split_dataset.npz'), allow_pickle=False)
    meta = json.load(open(os.path.join(here, 'g12_split_dataset.json')))[f'{name}/{split}']
    ds = S.load_split_dataset(os.path.join(here, name), split,
                              user_feature_definitions=[{'name': 'gender', 'type': 'categorical'}, {'name': 'age', 'type': 'discrete'}],
                              item_feature_definitions=[{'name': 'genres', 'type': 'tag', 'tag_split_sep': '|'},
                                                        {'name': 'text', 'type': 'vector'}], n_negative_samples=3)
    p = f'{name}/{split}'
    assert (ds.n_users, ds.n_items, ds.n_interactions) == (meta['n_users'], meta['n_items'], meta['n_interactions'])
    assert (ds.is_cold_start_user, ds.is_cold_start_item) == (meta['is_cold_start_user'], meta['is_cold_start_item'])
    assert np.array_equal(ds.users_in_split, z[p + '/users_in_split']) and np.array_equal(ds.items_in_split, z[p + '/items_in_split'])
    for mname in ('interaction_matrix', 'user_sampling_matrix', 'user_sampling_matrix_train', 'item_sampling_matrix_train'):
        _csr_eq(getattr(ds, mname), z, f'{p}/{mname}')
    if split != 'train':
        _csr_eq(ds.exclude_data, z, f'{p}/exclude_data')
    for ent, feats in (('user', ds.user_features), ('item', ds.item_features)):
        for fname, f in feats.items():
            key, fm = f'{p}/{ent}/{fname}', meta['features'][f'{ent}/{fname}']
            assert np.array_equal(np.asarray(f._indices), z[key + '/indices']), key
            want = z[key + '/values']
            if fm['type'] == 'tag':
                # same tag ids per row; the order inside a row is the reference's set order there, sorted here
                pad = len(fm['unique_values'])
                assert f.dim == fm['dim'] == pad
                got = np.asarray(f.values)
                assert got.shape[0] == want.shape[0]
                for a, b in zip(got, want):
                    assert sorted(int(v) for v in a if v != pad) == sorted(int(v) for v in b if v != pad)
            elif fm['type'] == 'categorical':
                assert f.n_unique_categories == len(fm['unique_values']) and np.array_equal(np.asarray(f.values), want)
            else:
                got = np.asarray(f.values, dtype=np.float64).reshape(len(want), -1)
                assert np.allclose(got, np.asarray(want, dtype=np.float64).reshape(len(want), -1))


@pytest.mark.parametrize('choice_set', ['all', 'subset'])
def test_dataset_level_samplers_match_reference_streams(choice_set):
    """Product-side dataset-level samplers (sampling.dataset_negative_*; a20 (ii) + (iv)) == the reference's streams (g13:
    values and generator position), and the loader's per-interaction mode == the oracle's literal collate."""
    from importlib import import_module
    import sibrar_amd as S
    from oracle import sampling_ref
    sampling = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.sampling')
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g13_dataset_samplers.npz'))
    w = world(load('g6_neg_sampling'))
    inter = w['inter']
    choices = z[f'{choice_set}/choices']
    pos = [np.intersect1d(inter[u].indices, choices) for u in range(10)]
    pop = z['pop_distribution']
    for name, fn in (('uniform', lambda u: sampling.dataset_negative_uniform(choices, 3, pos[u])),
                     ('uniform_recbole', lambda u: sampling.dataset_negative_uniform_recbole(choices, 3, pos[u])),
                     ('popular_1.0', lambda u: sampling.dataset_negative_popular(choices, 3, pop, 1.0, pos[u])),
                     ('popular_0.75', lambda u: sampling.dataset_negative_popular(choices, 3, pop, 0.75, pos[u]))):
        np.random.seed(42)
        got = np.stack([fn(u) for u in range(10)])
        assert np.array_equal(got, z[f'{choice_set}/{name}']), name
        assert np.array_equal(np.random.randint(0, 1000, size=4), z[f'{choice_set}/after_{name}']), name
    if choice_set == 'subset':
        return
    ds = S.SyntheticDataset.__new__(S.SyntheticDataset)
    ds.interaction_matrix = inter.tocoo()
    ds.user_sampling_matrix = inter
    ds.items_in_split = np.arange(I)
    ds.n_items = I
    ds.n_negative_samples = 3
    ds.sampling_popularity_squashing_factor = 0.75
    positives = [inter[u].indices for u in range(U)]
    for strategy in ('uniform', 'uniform_recbole', 'popular'):
        ds.negative_sampling_strategy = strategy
        sampling_ref.reproducible(42)
        loader = S.NegativeSamplingDataLoader(ds, batch_size=16, shuffle=True, use_dataset_negative_sampler=True)
        batches = [b for _, b in zip(range(3), loader)]
        sampling_ref.reproducible(42)
        perm = sampling_ref.loader_epoch_order(ds.interaction_matrix.nnz)
        for b, (u, i, l) in enumerate(batches):
            sel = perm[b * 16:(b + 1) * 16]
            ru, ri, rl = sampling_ref.dataset_sampler_collate(ds.interaction_matrix.row[sel], ds.interaction_matrix.col[sel], 3, strategy,
                                                              np.arange(I), positives, pop, 0.75)
            assert np.array_equal(u.numpy(), ru) and np.array_equal(i.numpy(), ri) and np.array_equal(l.numpy(), rl), strategy


@pytest.mark.parametrize('B,n_neg,subset', [(16, 3, False), (256, 10, False), (64, 5, True), (1, 1, False), (300, 7, True)])
def test_native_small_batch_collate_matches_the_numpy_formulation(B, n_neg, subset, monkeypatch):
    """sbr_host_recbole_collate (one native call for the whole default collate of a small batch) == the numpy formulation of
    recbole_negative_collate: same items, same labels, same generator state afterwards — on a dense world where most first
    draws collide (many redraw rounds), with an identity and a gapped item set."""
    from importlib import import_module
    import scipy.sparse as sp
    import sibrar_amd as S
    sampling = import_module(S.ops.__name__.rsplit('.', 1)[0] + '.sampling')
    rng = np.random.default_rng(B * 31 + n_neg)
    n_users, n_items = 40, 60
    inter = sp.csr_matrix((rng.random((n_users, n_items)) < 0.6).astype(np.int8))       # 60 % of all pairs are positives
    inter.sort_indices()
    items_in_split = np.sort(rng.choice(n_items, size=45, replace=False)) if subset else np.arange(n_items)
    users = rng.integers(0, n_users, size=B)
    pos_items = rng.integers(0, n_items, size=B)
    index = sampling.PositiveIndex(inter)
    outs = []
    for native in (False, True):
        monkeypatch.setenv('SBR_NATIVE_COLLATE', '1' if native else '0')
        np.random.seed(123)
        np.random.randint(0, 9, size=B % 17)                     # arbitrary position in the stream
        u, i, l = sampling.recbole_negative_collate(users, pos_items, n_neg, items_in_split, index)
        outs.append((u, i, l, np.random.randint(0, 1000, size=6)))
    for a, b in zip(outs[0], outs[1]):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert not any(inter[u, v] for u, row in zip(outs[1][0], outs[1][1]) for v in row[1:])      # no negative is a positive


def test_natural_key_order_of_metric_dictionaries():
    """FullEvaluator.get_results orders its keys like the reference's ``natsorted`` (eval/eval.py:160): cut-offs compare as numbers."""
    from importlib import import_module
    ev = import_module('sibrar---single-branch-recommender_amd.evaluation')
    from oracle import eval_ref
    keys = ['ndcg@10', 'ndcg@3', 'ndcg@100', 'ndcg@10_std', 'val/recall@20', 'val/recall@5', 'coverage@1', 'gender_f/ndcg@10',
            'gender_m/ndcg@2', 'precision@1', 'f_score@50']
    got = sorted(keys, key=ev.natural_key)
    assert got == eval_ref.natural_sorted(keys)
    assert got.index('ndcg@3') < got.index('ndcg@10') < got.index('ndcg@10_std') < got.index('ndcg@100')
    assert got.index('val/recall@5') < got.index('val/recall@20')


def test_full_evaluator_config_validation():
    """Group metrics need categorical user features that exist (eval/eval.py:76-92); unsupported metrics raise (:41-43)."""
    import sibrar_amd as S
    from types import SimpleNamespace
    from importlib import import_module
    ev = import_module('sibrar---single-branch-recommender_amd.evaluation')
    feats = {'gender': S.HostFeature('gender', 'categorical', np.array([0, 1, 1]), unique_values=['F', 'M']),
             'vec': S.HostFeature('vec', 'dense', np.zeros((3, 2), np.float32))}
    ds = SimpleNamespace(user_features=feats)
    assert S.FullEvaluator(ev._Cfg(calculate_group_metrics=True), dataset=ds)._user_features == ['gender']
    assert S.FullEvaluator(ev._Cfg(calculate_group_metrics=True, user_group_features=['gender']), dataset=ds)._user_features == ['gender']
    with pytest.raises(ValueError, match='does not contain user feature'):
        S.FullEvaluator(ev._Cfg(calculate_group_metrics=True, user_group_features=['age']), dataset=ds)
    with pytest.raises(ValueError, match='is not categorical'):
        S.FullEvaluator(ev._Cfg(calculate_group_metrics=True, user_group_features=['vec']), dataset=ds)
    with pytest.raises(ValueError, match='not supported'):
        S.FullEvaluator(ev._Cfg(metrics=['ndcg', 'auc']))
    assert list(feats['gender'].get_labels([1, 0])) == ['M', 'F']


def _pcg_state_words(rng):
    st = rng.bit_generator.state
    s, inc = int(st['state']['state']), int(st['state']['inc'])
    m = (1 << 64) - 1
    return [s >> 64, s & m, inc >> 64, inc & m, int(st['has_uint32']), int(st['uinteger'])]


def test_native_pcg64_modality_draw_equals_numpy_generator():
    """csrc/producer.hip's PCG64 + Lemire replica against numpy's Generator through the product's Python formulation
    (sampling.sample_modalities, itself pinned by the G7 golden streams): k = 1, k = 2 and central-modality draws for 1..6
    modalities, from fresh generators and from states with a buffered 32-bit half; positions, per-modality counts and the
    generator state afterwards must be identical (checked by drawing once more from both)."""
    import ctypes
    import sibrar_amd as S
    from importlib import import_module
    _lib = import_module('sibrar---single-branch-recommender_amd._lib')
    lib = _lib.lib()
    for seed in (42, 7):
        for n_mod in (1, 2, 3, 4, 6):
            for reg, k, central in (('no_regularization', 1, None), ('pairwise_single', 2, None), ('central_modality', 2, 1)):
                if k > n_mod or (central is not None and n_mod < 2):
                    continue
                order = [f'm{i}' for i in range(n_mod)]
                for warm in (0, 1, 3):                      # odd numbers of 32-bit draws leave a buffered half behind
                    ref = np.random.default_rng(seed)
                    for _ in range(warm):
                        ref.integers(0, 5)
                    words = (ctypes.c_ulonglong * 6)(*_pcg_state_words(ref))
                    n_slots = 2816
                    want = S.sampling.sample_modalities(ref, order, n_slots, reg, order[central] if central is not None else None)
                    got = np.empty((n_slots, k), dtype=np.int8)
                    counts = (ctypes.c_long * 8)()
                    rc = lib.sbr_host_pcg64_modalities(ctypes.cast(words, ctypes.c_void_p), n_slots, n_mod, k,
                                                       -1 if central is None else central, got.ctypes.data,
                                                       ctypes.cast(counts, ctypes.c_void_p))
                    assert rc == 0, lib.sbr_last_error()
                    assert np.array_equal(got, want), (seed, n_mod, reg, warm)
                    assert list(counts)[:n_mod] == np.bincount(want.reshape(-1), minlength=n_mod).tolist()
                    assert list(words) == _pcg_state_words(ref), 'generator state after the draw'


def test_native_plan_padding_equals_engine_plan():
    """sbr_host_pad_counts == the bucket arithmetic of engine._EntityRun.plan (graph-mode capacities of the slot lists)."""
    import ctypes
    import math
    from importlib import import_module
    lib = import_module('sibrar---single-branch-recommender_amd._lib').lib()
    rng = np.random.default_rng(0)
    for R in (6, 64, 2816, 5632, 90112, 180224, 999, 123457):
        for n_mod in (1, 2, 3, 5):
            raw = rng.multinomial(R, np.ones(n_mod) / n_mod)
            raw[rng.integers(0, n_mod)] = 0 if n_mod > 1 else raw[0]
            bucket = max(64, -(-int(5.0 * math.sqrt(R)) // 64) * 64)
            off = (R // n_mod + bucket // 2) % bucket
            want = np.where(raw > 0, (np.maximum(raw - off, 0) + bucket - 1) // bucket * bucket + off, 0)
            got = (ctypes.c_long * n_mod)(*[int(c) for c in raw])
            assert lib.sbr_host_pad_counts(ctypes.cast(got, ctypes.c_void_p), n_mod, R) == 0
            assert list(got) == want.tolist(), (R, n_mod)


@pytest.mark.parametrize('name', ['split_random', 'split_cold_item'])
def test_device_containers_built_from_reference_objects_equal_the_product_loader(name):
    """Boundary evidence (g16, tests/golden/make_golden_boundary.py): ``DeviceTable`` / ``_csr_to_device`` / ``FullEvaluator``
    were fed the reference's REAL ``Feature`` and ``FullEvalDataset`` objects in the build container and every buffer they
    would upload was recorded. Here the same buffers are built from the product's own loader (load_split_dataset ->
    HostFeature) and from the HostFeatures SingleBranchNet adds itself: layouts, id -> row maps, category folds, padded tag
    matrices (as tag sets per row: the reference pads in set order), CSR arrays, exclusion and label CSRs are identical."""
    import json
    import sibrar_amd as S
    from importlib import import_module
    evaluation = import_module('sibrar---single-branch-recommender_amd.evaluation')
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'g16_boundary.npz'), allow_pickle=False)
    meta = json.load(open(os.path.join(here, 'g16_boundary.json')))
    fdefs = dict(user_feature_definitions=[{'name': 'gender', 'type': 'categorical'}, {'name': 'age', 'type': 'discrete'}],
                 item_feature_definitions=[{'name': 'genres', 'type': 'tag', 'tag_split_sep': '|'}, {'name': 'text', 'type': 'vector'}])
    n_tables = 0
    for split in ('train', 'val', 'test'):
        ds = S.load_split_dataset(os.path.join(here, name), split, n_negative_samples=3, **fdefs)
        feats = {('user', k): f for k, f in ds.user_features.items()}
        feats.update({('item', k): f for k, f in ds.item_features.items()})
        feats[('user', 'interactions')] = S.HostFeature('interactions', 'csr', ds.user_sampling_matrix_train)
        feats[('user', 'user_embedding')] = S.HostFeature('user_embedding', 'categorical', np.arange(ds.n_users), n_categories=ds.n_users)
        feats[('item', 'interactions')] = S.HostFeature('interactions', 'csr', ds.item_sampling_matrix_train)
        feats[('item', 'item_embedding')] = S.HostFeature('item_embedding', 'categorical', np.arange(ds.n_items), n_categories=ds.n_items)
        for (ent, k), f in feats.items():
            p = f'{name}/{split}/{ent}/{k}'
            t, m = S.DeviceTable(f), meta[p]
            assert (t.kind, int(t.dim), int(t.n_rows)) == (m['kind'], m['dim'], m['n_rows']), p
            for attr in ('n_categories', 'pad', 'T', 'binary'):
                assert (attr in m) == hasattr(t, attr) and (attr not in m or int(getattr(t, attr)) == m[attr]), (p, attr)
            for buf in ('values', 'tags', 'indptr', 'indices', 'data', 'rowmap'):
                got = getattr(t, buf, None)
                assert (got is not None) == (f'{p}/{buf}' in z.files), (p, buf)
                if got is None:
                    continue
                want = z[f'{p}/{buf}']
                if buf == 'tags':
                    assert got.shape == want.shape
                    for a, b in zip(got.numpy(), want):
                        assert sorted(a.tolist()) == sorted(b.tolist()), p
                elif got.dtype.is_floating_point:
                    assert np.allclose(got.numpy(), want), (p, buf)
                else:
                    assert np.array_equal(got.numpy(), want), (p, buf)
            n_tables += 1
        if split != 'train':
            p = f'{name}/{split}'
            ip, ix = evaluation._csr_to_device(ds.exclude_data, 'cpu')
            assert np.array_equal(ip.numpy(), z[p + '/exclude/indptr']) and np.array_equal(ix.numpy(), z[p + '/exclude/indices'])
            lp, lx = S.FullEvaluator(dataset=ds)._labels('cpu')
            assert np.array_equal(lp.numpy(), z[p + '/labels/indptr']) and np.array_equal(lx.numpy(), z[p + '/labels/indices'])
            assert np.array_equal(np.asarray(ds.items_in_split), z[p + '/items_in_split'])
            assert np.array_equal(np.asarray(ds.users_in_split), z[p + '/users_in_split'])
    assert n_tables == 24


@pytest.mark.parametrize('kind', ['csr', 'tag'])
def test_transposed_feature_tables_equal_the_scipy_transpose(kind):
    """features.DeviceTable.transposed — the operand of the gather-form weight gradient of the CSR projector and of the tag bag
    (sbr_csr_project_bwd_gather): the CSR arrays of X^T for the 'interactions' matrix (data/Feature.py:149-150), and of the transpose
    of X[entity, tag] = 1 / (tags of the entity) for a padded tag list (nn.EmbeddingBag(mean, padding), sgd_alg.py:1336-1337), built
    with torch ops only (so on the table's device) == scipy's transpose: indptr, entity rows ascending within a column, values;
    weighted and all-ones matrices, empty rows and columns, an entity without tags, a duplicated tag."""
    import scipy.sparse as sp
    from importlib import import_module
    features = import_module('sibrar---single-branch-recommender_amd.features')
    rng = np.random.default_rng(3)
    n_ent, n_cols = 60, 23
    if kind == 'csr':
        for weighted in (False, True):
            m = sp.random(n_ent, n_cols, density=0.2, format='csr', random_state=4, dtype=np.float32)
            m.data[:] = rng.standard_normal(m.nnz).astype(np.float32) if weighted else 1.0
            m = m.tolil(); m[7, :] = 0; m[:, 5] = 0; m = m.tocsr(); m.eliminate_zeros()
            t = features.DeviceTable(features.HostFeature('interactions', 'csr', m))
            ip, ix, dv = t.transposed()
            ref = m.T.tocsr(); ref.sort_indices()
            assert np.array_equal(ip.numpy(), ref.indptr) and np.array_equal(ix.numpy(), ref.indices)
            assert (dv is None) == (not weighted)
            if weighted:
                assert np.array_equal(dv.numpy(), ref.data)
            assert t.transposed()[0] is ip                                     # built once
    else:
        T = 5
        tags = np.full((n_ent, T), n_cols, dtype=np.int64)
        for e in range(n_ent):
            k_ = int(rng.integers(0, T + 1)) if e != 9 else 0
            tags[e, :k_] = rng.choice(n_cols, size=k_, replace=False)
        tags[3, :3] = [4, 4, 6]                                                # a duplicated tag counts twice, as in the bag's mean
        t = features.DeviceTable(features.HostFeature('genres', 'tag', tags, n_categories=n_cols))
        ip, ix, dv = t.transposed(n_cols + 1)
        dense = np.zeros((n_ent, n_cols + 1), dtype=np.float64)
        for e in range(n_ent):
            real = tags[e][tags[e] != n_cols]
            for g_ in real:
                dense[e, g_] += 1.0 / len(real)
        got = sp.csr_matrix((dv.numpy().astype(np.float64), ix.numpy(), ip.numpy()), shape=(n_cols + 1, n_ent)).toarray()
        assert np.allclose(got, dense.T, rtol=1e-6, atol=0) and ip.numel() == n_cols + 2 and int(ip[-1]) == int((tags != n_cols).sum())
        for c in range(n_cols + 1):                                            # entity rows ascending within a tag
            seg = ix.numpy()[int(ip[c]):int(ip[c + 1])]
            assert np.all(np.diff(seg) >= 0)
