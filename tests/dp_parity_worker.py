"""One rank of the N-rank numerical parity test (tests/test_hip_pinned.py::test_two_rank_data_parallel_parity_with_the_oracle):
three fused data-parallel steps from a golden state_dict on this rank's [rank::world] shard of the golden batches and recorded
modality decisions; rank 0 writes its final state_dict. Ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks
on one device). Usage: python dp_parity_worker.py <case> <rank> <world> <rendezvous file> <out.npz>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    case_name, rank, world, rdzv, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    from golden_util import MANIFEST, I, load, product_net
    import sibrar_amd as S
    dist.init_process_group('gloo', init_method=f'file://{rdzv}', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    z = load('g8_optim')
    case = [c for c in MANIFEST['g8_optim']['cases'] if c['name'] == case_name][0]
    net = product_net(z, case, f'{case_name}/sd0/')
    net.train()
    opt = S.FusedOptimizer(net, case['optimizer'], lr=case['lr'], weight_decay=case['wd'])
    loss_fn = S.RecBayesianPersonalizedRankingLoss(n_items=I, aggregator='mean', train_neg_strategy='uniform_recbole', neg_train=3)
    fused = S.FusedTrainStep(net, loss_fn, opt)
    assert fused.split and S.parallel.is_distributed()

    def draw_of(ent, names):
        order = ent.train_modality_order
        lut = {m: i for i, m in enumerate(order)}
        return np.vectorize(lut.__getitem__, otypes=[np.int8])(np.asarray(names)).reshape(-1, names.shape[-1]), order

    losses, seen = [], []
    real_step = opt.step_flat

    def recording_step(*a, **k):                      # the averaged gradients every replica is about to apply
        seen.append({name: p.grad.detach().cpu().numpy().copy() for name, p in net.named_parameters()})
        return real_step(*a, **k)
    opt.step_flat = recording_step
    for s in range(3):
        u, i, labels = (torch.from_numpy(z[f'{case_name}/{k}{s}'])[rank::world].contiguous() for k in ('u', 'i', 'labels'))
        key = f'{case_name}/user_mods{s}'
        du = draw_of(net.user_embedding_module, z[key][rank::world]) if key in z.files else None
        di = draw_of(net.item_embedding_module, z[f'{case_name}/item_mods{s}'][rank::world])
        total, rec, reg = fused.step(u, i, labels, (du, di))
        losses.append(float(rec))
    torch.cuda.synchronize()
    exchange = 'sparse' if fused._sparse else 'dense'
    fused.close()
    grads = {f'g{s}/{k}': v for s, d in enumerate(seen) for k, v in d.items()}
    np.savez(out_path + f'.rank{rank}.npz', losses=np.array(losses), exchange=np.array(exchange), **grads,
             **{k: v.detach().cpu().numpy() for k, v in net.state_dict().items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
