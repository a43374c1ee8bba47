"""Line-level timing of one collate + prepare at B=8192 on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sibrar_amd as S
import bench
dev = 'cuda:0'
torch.set_num_threads(bench.host_cores())
ds, net = bench.build(S, dict(bench.C2), dev)
B, n_neg = 8192, 10
coo = ds.interaction_matrix
rows, cols = coo.row.astype(np.int64), coo.col.astype(np.int64)
pos = S.sampling.DevicePositiveIndex(ds.user_sampling_matrix, dev)
order = np.random.default_rng(0).permutation(len(rows))
T = {}


def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


N = 50
for it in range(N + 5):
    if it == 5:
        T.clear()
    t = time.perf_counter()
    sel = order[it * B:(it + 1) * B]
    u, p = rows[sel], cols[sel]
    t = tick('gather rows/cols', t)
    slot_user = np.tile(u, n_neg)
    t = tick('tile', t)
    values = np.random.randint(0, ds.n_items, size=B * n_neg)
    t = tick('randint', t)
    hit = pos.contains(slot_user, values)
    t = tick('contains round 1', t)
    todo = np.flatnonzero(hit)
    t = tick('flatnonzero', t)
    while len(todo):
        values[todo] = np.random.randint(0, ds.n_items, size=len(todo))
        todo = todo[pos.contains(slot_user[todo], values[todo])]
    t = tick('redraw rounds', t)
    items = np.empty((B, 1 + n_neg), dtype=np.int64)
    items[:, 0] = p
    items[:, 1:] = values.reshape(n_neg, B).T
    t = tick('assemble items', t)
    labels = np.zeros(items.shape)
    labels[:, 0] = 1
    t = tick('labels', t)
print('--- collate, ms per batch')
for k, v in T.items():
    print(f'{k:22s} {v / N * 1e3:7.3f}')
print(f'{"total":22s} {sum(T.values()) / N * 1e3:7.3f}')

# inside contains
T.clear()
users = np.tile(rows[order[:B]], n_neg); vals = np.random.randint(0, ds.n_items, size=B * n_neg)
for it in range(N):
    t = time.perf_counter()
    with torch.cuda.stream(pos.stream):
        a = np.ascontiguousarray(users, dtype=np.int64); b = np.ascontiguousarray(vals, dtype=np.int64)
        t = tick('ascontig', t)
        ut = torch.from_numpy(a).to(dev, non_blocking=True); vt = torch.from_numpy(b).to(dev, non_blocking=True)
        t = tick('h2d x2 (pageable)', t)
        out = torch.empty(len(users), dtype=torch.uint8, device=dev)
        S._lib.call('sbr_csr_contains', S._lib.ptr(pos.indptr), S._lib.ptr(pos.indices), S._lib.ptr(ut), S._lib.ptr(vt), len(users), S._lib.ptr(out), pos.stream.cuda_stream) if hasattr(S, '_lib') else None
        t = tick('launch', t)
        res = out.cpu()
        t = tick('d2h + sync', t)
        r = res.numpy().astype(bool)
        t = tick('astype', t)
print('--- contains(81920), ms')
for k, v in T.items():
    print(f'{k:22s} {v / N * 1e3:7.3f}')
