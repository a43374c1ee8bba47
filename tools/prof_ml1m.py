"""Where the host time goes for the shipped ML-1M config shape at B=256 (collate rounds) and in evaluation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
sys.argv = [sys.argv[0], '__none__']
dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())
ds = S.SyntheticDataset(5816, 3299, 651034, item_dense={'plot_mpnet': 768}, item_tags={'genres': (18, 3)},
                        user_categorical={'gender': 2, 'occupation': 21}, seed=0, n_negative_samples=10, holdout_per_user=2)
ld = S.NegativeSamplingDataLoader(ds, batch_size=256, shuffle=True, device=dev, prefetch=0)
it = iter(ld)
for _ in range(5):
    next(it)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(100):
    next(it)
print(f'collate {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
