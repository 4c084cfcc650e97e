#!/usr/bin/env python
"""Does the placement of dx relative to x and g matter?  The stats-scaled backward on the headline tensor with dx carved
out of a larger allocation at a byte offset (16-byte multiples), interleaved rounds, HIP events.

    python tools/dx_offset_probe.py [--offsets 0,256,4096,...]
"""
import argparse
import statistics
import sys

import torch

sys.path.insert(0, '.')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='256,512,56,56')
    ap.add_argument('--offsets', default='0,256,1024,4096,65536,1048576,1052672,3145728,8388608,8392704')
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    from brevitas_amd import _native as nat
    n, c, h, w = (int(v) for v in args.shape.split(','))
    dt = torch.bfloat16
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    x = torch.randn(n, c, h, w, device=dev, dtype=dt).reshape(-1)
    g = torch.randn(n, c, h, w, device=dev, dtype=dt).reshape(-1)
    big = torch.empty(x.numel() + (16 << 20), device=dev, dtype=dt)   # dx lives somewhere in here
    code = nat.dtype_code(dt)
    d = nat.QuantDesc(n, c, h * w, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, nat.OUT_DEQUANT, 0)
    zp = torch.zeros(1, device=dev)
    stat, scale = nat.absmax_scale(x, n, c, h * w, 1e-10, 128.0, dt)
    print('x %#x  g %#x  big %#x  (tensor bytes %#x)' % (x.data_ptr(), g.data_ptr(), big.data_ptr(), x.numel() * 2))
    real_empty_like = torch.empty_like
    offs = [int(v) for v in args.offsets.split(',')]
    res = {o: [] for o in offs}

    def timed(off):
        def fake(t, *a, **k):
            if t is x:
                return big[off // 2: off // 2 + x.numel()]
            return real_empty_like(t, *a, **k)
        nat.torch.empty_like = fake
        try:
            nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.iters
        finally:
            nat.torch.empty_like = real_empty_like

    for _ in range(args.rounds):
        for o in offs:
            res[o].append(timed(o))
    for o in offs:
        v = res[o]
        print('dx offset %9d B (dx - x = %#x mod 16 MiB: %#x)  median %.4f ms  min %.4f' % (
            o, big.data_ptr() + o - x.data_ptr(), (big.data_ptr() + o - x.data_ptr()) % (16 << 20),
            statistics.median(v), min(v)))


if __name__ == '__main__':
    main()
