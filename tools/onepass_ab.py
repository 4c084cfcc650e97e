#!/usr/bin/env python
"""Interleaved A/B rounds, one process: the one-launch routes (last-arriving wave finishes the channel) against the
two-launch routes, on the headline tensor.  HIP events around each call, medians over rounds.

    python tools/onepass_ab.py [--shape 256,512,56,56] [--rounds 12] [--iters 20]
"""
import argparse
import statistics
import sys

import torch

sys.path.insert(0, '.')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='256,512,56,56')
    ap.add_argument('--rounds', type=int, default=12)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--dtype', default='bf16')
    args = ap.parse_args()
    from brevitas_amd import _native as nat
    n, c, h, w = (int(v) for v in args.shape.split(','))
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    x = torch.randn(n, c, h, w, device=dev, dtype=dt)
    g = torch.randn(n, c, h, w, device=dev, dtype=dt)
    flat, gflat = x.reshape(-1), g.reshape(-1)
    inner = h * w
    code = nat.dtype_code(dt)
    d = nat.QuantDesc(n, c, inner, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, nat.OUT_DEQUANT, 0)
    zp = torch.zeros(1, device=dev)
    run = torch.ones(c, device=dev, dtype=dt)
    stat, scale = nat.absmax_scale(flat, n, c, inner, 1e-10, 128.0, dt)

    def stat_call():
        nat.absmax_scale(flat, n, c, inner, 1e-10, 128.0, dt, 0, running=run, momentum=0.1, first_batch=False)

    def bwd_call():
        nat.fakequant_bwd_stats(d, gflat, flat, scale, zp, stat, dt, 128.0, dt)

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters

    res = {}
    for _ in range(3):  # settle
        stat_call()
        bwd_call()
    for r in range(args.rounds):
        for name, fn, attr in (('absmax', stat_call, 'ONEPASS'), ('backward', bwd_call, 'ONEPASS_BWD')):
            for on in (True, False):
                setattr(nat, attr, on)
                res.setdefault((name, on), []).append(timed(fn))
        nat.ONEPASS = nat.ONEPASS_BWD = True
    b = x.element_size()
    for name, passes in (('absmax', 1), ('backward', 3)):
        for on in (True, False):
            v = res[(name, on)]
            med = statistics.median(v)
            print('%-9s %-10s median %.4f ms  min %.4f  max %.4f   %.2f TB/s' % (
                name, 'one-launch' if on else 'two-launch', med, min(v), max(v), passes * b * x.numel() / med / 1e9))


if __name__ == '__main__':
    main()
