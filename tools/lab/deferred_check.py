"""Debug aid: run-to-run reproducibility of the fused step on a tiny world (2000 users, batches of 48), dense optimizer, plain
launches. Variant a: every 5th batch draws its users from 60 ids (several slots per user: the float atomics of the table gradient
add in varying order) -> two outcomes, 0.08 apart in one parameter, the rarer one in ~5 % of runs (a last-bit difference flips a
ReLU gate / the sign of a near-zero Adam step). Variants b (users from all 2000 ids) and c (distinct users per batch): bit-identical
runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sibrar_amd as S
DEV = 'cuda'
ds = S.SyntheticDataset(2000, 200, 9000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
       'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
HIST = {}
VARIANT = 'a'
def run(deferred, graph, steps=25, mid=True, dup=True):
    os.environ['SBR_DEFERRED_ADAM'] = deferred
    torch.manual_seed(11); np.random.seed(11)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV); net.train()
    opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
    loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S.FusedTrainStep(net, loss, opt, use_graph=graph)
    rng = np.random.default_rng(9)
    for s_ in range(steps):
        if VARIANT == 'a':
            u = torch.from_numpy(rng.integers(0, 60 if s_ % 5 == 0 else ds.n_users, size=48)); u[1] = u[0]
        elif VARIANT == 'b':
            u = torch.from_numpy(rng.integers(0, ds.n_users, size=48))
        else:
            u = torch.from_numpy(rng.permutation(60 if s_ % 5 == 0 else ds.n_users)[:48].copy())
        HIST[s_] = u.tolist()
        i = torch.from_numpy(rng.integers(0, ds.n_items, size=(48, 4)))
        labels = torch.zeros(48, 4, dtype=torch.float64); labels[:, 0] = 1
        fused.step(u, i, labels)
        if mid and s_ == 12: net.state_dict()
    fused.close()
    return net.state_dict()['user_embedding_module.embedding_layer.weight'].cpu().clone()
os.environ['SBR_DEFERRED_SYNC'] = ''; os.environ['SBR_DEFERRED_PRESYNC'] = ''
for VARIANT in ('a', 'b', 'c'):
    ref = run('0', False, mid=False)
    outcomes = {}
    for rep in range(80):
        got = run('0', False, mid=False)
        d = round(float((got - ref).abs().max()), 6)
        key = d if d > 1e-3 else 0.0
        outcomes[key] = outcomes.get(key, 0) + 1
    print(f'variant {VARIANT} (dense optimizer, plain launches): max-diff outcomes over 80 runs {outcomes}', flush=True)
