"""Sustained time per training step of the bench's c2 model through the bench's own loader pipeline, with the replay count of the
captured step (a step that is not replayed runs ~50 plain launches).   usage: python tools/lab/step_time.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import sibrar_amd as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = 'cuda:0'
if os.environ.get('SPLIT_MIN_ROWS'):       # lab: row threshold of the bf16-split GEMM kernels (ops._SPLIT_MIN_ROWS)
    S.ops._SPLIT_MIN_ROWS = int(os.environ['SPLIT_MIN_ROWS'])
ds, net = bench.build(S, bench.C2, dev)
loss = S.RecSampledSoftmaxLoss(n_items=ds.n_items, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=ds.n_negative_samples)
trainer = S.Trainer(net, None, None, loss, bench._Conf(dev))
net.train()
np.random.seed(42)
loader = S.NegativeSamplingDataLoader(ds, batch_size=B, shuffle=True, device=dev, dp_sampling='local', prefetch=4,
                                      prepare_fn=trainer.fused.prepare)
it = bench.epochs(loader)
if os.environ.get('SIDE'):                      # lab: a second stream that has been used once, as a previous trainer would leave it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.zeros(8, device=dev).add_(1)
    torch.cuda.current_stream().wait_stream(side)
    if os.environ['SIDE'] == '2':
        del side
bench.run_steps(S, trainer, it, 35, 1)
torch.cuda.synchronize()
print('after warm-up: replays', trainer.fused.n_replays, 'graphs', {k[0]: type(v).__name__ for k, v in trainer.fused._graphs.items()}, flush=True)
t0 = time.perf_counter()
bench.run_steps(S, trainer, it, STEPS, 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'B={B}: {1e3 * dt / STEPS:.3f} ms per step, replays {trainer.fused.n_replays} of {trainer.fused.n_steps} steps')
loader.close(); trainer.fused.close()
