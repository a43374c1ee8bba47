"""Debug aid: dense vs dense vs deferred Adam on the same stream of batches (max abs differences of the user table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sibrar_amd as S
DEV = 'cuda'
ds = S.SyntheticDataset(2000, 200, 9000, item_dense={'text': 40}, seed=3, n_negative_samples=3)
cfg = {'shared_common_dim': 32, 'user': {'feature_name': 'user_embedding', 'embedding_dim': -1},
       'item': {'features': [{'feature_name': 'text'}, {'feature_name': 'item_embedding'}],
                'single_branch_hidden_layers': [32], 'preference_hidden_layers': [], 'common_modality_dim': 32}}
def run(deferred, steps=25, dup=True):
    os.environ['SBR_DEFERRED_ADAM'] = deferred
    torch.manual_seed(11); np.random.seed(11)
    net = S.SingleBranchNet(S.SingleBranchNetConfig.from_dict(cfg), ds).to(DEV); net.train()
    opt = S.FusedOptimizer(net, 'adamw', lr=1e-2, weight_decay=1e-2)
    loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S.FusedTrainStep(net, loss, opt, use_graph=False)
    rng = np.random.default_rng(9)
    for s_ in range(steps):
        u = torch.from_numpy(rng.integers(0, 60 if s_ % 5 == 0 else ds.n_users, size=48))
        if dup: u[1] = u[0]
        i = torch.from_numpy(rng.integers(0, ds.n_items, size=(48, 4)))
        labels = torch.zeros(48, 4, dtype=torch.float64); labels[:, 0] = 1
        fused.step(u, i, labels)
    fused.close()
    return net.state_dict()['user_embedding_module.embedding_layer.weight'].cpu().clone()
for steps in (1, 2, 3, 25):
    for dup in (False, True):
        a, b, c = run('0', steps, dup), run('0', steps, dup), run('1', steps, dup)
        print(f'steps {steps} dup {dup}: dense-dense {float((a-b).abs().max()):.3e}  dense-deferred {float((a-c).abs().max()):.3e}  rows differing {(a!=c).any(1).sum().item()}')
