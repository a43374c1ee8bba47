#!/usr/bin/env python3
"""Build-time check of the last-arriver hand-offs (gemm_split_kernel<0,2>, bn_score_loss_kernel, rec_loss_kernel<3>).

Every arrival-counter add (the only RETURNING 64-bit atomic add of those kernels: ``global_atomic_add_x2 vDST, ... sc0``) must be
preceded — with no vector-memory instruction of its own wave in between — by ``s_waitcnt vmcnt(0)``: the data the last arriver
reads (column-sum atomics, loss partials) has then been performed before the counter moves (MI355X_MICROARCH.md, Valid forms).
And between the last data atomic / store of the kernel and that counter add there must be such a wait in front of the workgroup
barrier. The compiler does not emit either wait for a workgroup-scope fence; the source uses SBR_DRAIN_VMEM() (inline asm). This
script compiles the three files device-only and fails loudly if a compiler or source change loses the waits.

    python tools/check_arrival_waits.py            # exit code 0 = all hand-offs drained
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'sibrar---single-branch-recommender_amd', 'csrc')
FILES = {'gemm_split_f32.hip': ['gemm_split_kernelILi0ELi2E'], 'fused_tail.hip': ['bn_score_loss_kernel'],
         'loss.hip': ['rec_loss_kernelILi3E']}
VMEM = re.compile(r'^\s*(global_|buffer_|flat_|scratch_)')
COUNTER = re.compile(r'^\s*global_atomic_add_x2\s+v\[\d+:\d+\],.*\bsc0\b')          # returning form: has a destination register pair
DATA = re.compile(r'^\s*global_atomic_add_f64\s')
BARRIER = re.compile(r'^\s*s_barrier')
DRAIN = re.compile(r'^\s*s_waitcnt\s+vmcnt\(0\)')


def kernels(asm):
    """-> {symbol: [instruction lines]} for every kernel body of the listing"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if line.strip().startswith('.end_amdhsa_kernel') or line.strip().startswith('s_endpgm'):
                cur.append(line)
                if line.strip().startswith('.end_amdhsa_kernel'):
                    cur = None
                continue
            cur.append(line)
    return out


def check(sym, body):
    adds = [i for i, l in enumerate(body) if COUNTER.match(l)]
    if not adds:
        return [f'{sym}: no returning 64-bit counter add found (pattern changed?)']
    errs = []
    for i in adds:
        j = i - 1
        seen_barrier_drain = False
        while j >= 0:
            l = body[j]
            if DRAIN.match(l):
                seen_barrier_drain = True
                break
            if VMEM.match(l):
                break
            j -= 1
        # kernels whose hand-off data are double atomics of ALL waves (column sums): behind the last of them a drain, THEN the
        # workgroup barrier, then the counter add
        data = [q for q in range(i) if DATA.match(body[q])]
        if data:
            between = body[data[-1] + 1:i]
            d = next((q for q, l in enumerate(between) if DRAIN.match(l)), None)
            b = next((q for q, l in enumerate(between) if BARRIER.match(l) and d is not None and q > d), None)
            if d is None or b is None:
                errs.append(f'{sym}: no "s_waitcnt vmcnt(0)" followed by s_barrier between the last column-sum atomic and the counter add')
        if not seen_barrier_drain:
            errs.append(f'{sym}: counter add at listing line {i} is not preceded by s_waitcnt vmcnt(0) (nearest VMEM: {body[j].strip() if j >= 0 else "none"})')
    return errs


def main():
    errs, n = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for f, wanted in FILES.items():
            out = os.path.join(tmp, f + '.s')
            subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-Wno-unused-command-line-argument',
                            os.path.join(CSRC, f), '-o', out], check=True, stderr=subprocess.DEVNULL)
            ks = kernels(open(out).read())
            for w in wanted:
                hits = [s for s in ks if w in s]
                if not hits:
                    errs.append(f'{f}: no kernel symbol containing {w}')
                for s in hits:
                    n += 1
                    errs += check(s, ks[s])
    if errs:
        print('\n'.join(errs))
        return 1
    print(f'arrival hand-offs drained in {n} kernels')
    return 0


if __name__ == '__main__':
    sys.exit(main())
