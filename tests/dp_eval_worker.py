"""One rank of the item-sharded evaluation test (tests/test_hip_pinned.py::test_two_rank_item_sharded_evaluation_equals_one_rank):
``evaluate_recommender_algorithm`` under an initialised process group — this rank computes the representations of its item shard
only, scores every user against it, the per-shard top-k lists are all-gathered and merged, every rank computes the metrics. Ranks
share cuda:0 and exchange over gloo (RCCL refuses two ranks on one device). Worlds: the golden G9 evaluation world (D = 8: fp32
route, also when the fused scorer is asked for) and a 3,000-user x 1,111-item world with D = 64 (fused kernel with item_offset; 1,111
items do not divide by the world size). Usage: python dp_eval_worker.py <rank> <world> <rendezvous file> <out prefix>"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def worlds(S, DEV):
    from golden_util import MANIFEST, I, U, load, product_net, world
    z = load('g9_eval')
    w = world(z)
    view = SimpleNamespace(n_users=U, n_items=I, items_in_split=np.arange(I), users_in_split=np.arange(U), n_items_in_split=I,
                           n_users_in_split=U, user_sampling_matrix=sp.csr_matrix(z['labels']), exclude_data=w['inter'].astype(bool))
    yield 'g9', product_net(z, MANIFEST['g9_eval'], 'sd/'), view, (1, 10, 20)
    ds = S.SyntheticDataset(3000, 1111, 40_000, item_dense={'text': 48}, item_tags={'genres': (12, 3)}, seed=5, n_negative_samples=5,
                            holdout_per_user=2)
    cfg = {'shared_common_dim': 64, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
           'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'genres'}, {'feature_name': 'item_embedding'}],
                    'single_branch_hidden_layers': [64], 'preference_hidden_layers': [], 'common_modality_dim': 64}}
    torch.manual_seed(5)
    np.random.seed(5)
    yield 'w64', S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV).eval(), ds.eval_view(), (1, 10, 20)


def run(S, DEV):
    out = {}
    for name, net, view, ks in worlds(S, DEV):
        for scorer in ('fp32', 'fp16_fused'):
            ev = S.FullEvaluator(config=S.evaluation._Cfg(top_k=ks, calculate_std=False), dataset=view)
            loader = type('L', (), {'dataset': view, 'batch_size': 64})()
            metrics, raw = S.evaluate_recommender_algorithm(net, loader, ev, DEV, return_raw=True, scorer=scorer, user_chunk=1024,
                                                            shard_items=True)
            for k, v in raw.items():
                out[f'{name}/{scorer}/{k}'] = np.asarray(v)
            for k, v in metrics.items():
                out[f'{name}/{scorer}/mean/{k}'] = np.float64(v)
    return out


def main():
    rank, world, rdzv, prefix = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import sibrar_amd as S
    if world > 1:
        dist.init_process_group('gloo', init_method=f'file://{rdzv}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    assert S.parallel.is_distributed() == (world > 1)
    res = run(S, 'cuda:0')
    torch.cuda.synchronize()
    np.savez(prefix + f'.w{world}.rank{rank}.npz', **res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
