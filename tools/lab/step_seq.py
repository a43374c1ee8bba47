"""Lab: B = 8192 steps, then B = 256 steps in the SAME process (as bench.py runs them): does one run leave something behind that
slows the next? (It did when the catch-up of the deferred table ran on a second stream: see DESIGN.md, round 3.)
usage: python tools/lab/step_seq.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import sibrar_amd as S
dev = 'cuda:0'
if os.environ.get('SWEEP') is not None:      # lab: sweep period of the deferred table (0: no sweep)
    from importlib import import_module
    import_module(S.ops.__name__.rsplit('.', 1)[0] + '.engine').DeferredTable.SWEEP_EVERY = int(os.environ['SWEEP'])
ds, net = bench.build(S, bench.C2, dev)
for B, n in ((8192, 100), (256, 400), (8192, 100), (256, 400)):
    dt, _ = bench.bench_training(S, ds, net, dev, B, n, 5, 0, 1, time_kernels=False)
    print(f'B={B}: {1e3 * dt / n:.3f} ms per step', flush=True)
