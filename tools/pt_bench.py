"""Developer tool: the per-tensor (one very long row) kernels on the headline tensor, per dtype -- used with the
BVQ_PIECE_CHUNKS / BVQ_MAX_UNITS_PER_CHANNEL knobs to pick the piece size of long rows.

    [BVQ_PIECE_CHUNKS=4 BVQ_MAX_UNITS_PER_CHANNEL=1048576] python tools/pt_bench.py [bf16,f32,f16]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402

DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}


def timed(fn, rounds=9):
    fn()
    fn()
    ts = []
    for _ in range(rounds):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['bf16', 'f32', 'f16']
    dev = 'cuda:0'
    n = 256 * 512 * 56 * 56
    print('# piece chunks %s, unit cap %s' % (os.environ.get('BVQ_PIECE_CHUNKS', 'default'),
                                             os.environ.get('BVQ_MAX_UNITS_PER_CHANNEL', 'default')))
    for _ in range(20):  # clock settling
        torch.empty(n, device=dev, dtype=torch.bfloat16).zero_()
    for name in names:
        dt = DT[name]
        x = torch.randn(n, device=dev, dtype=dt)
        g = torch.randn(n, device=dev, dtype=dt)
        b = x.element_size()
        code = nat.dtype_code(dt)
        d = nat.QuantDesc(1, 1, n, code, code, nat.F32, nat.F32, 0, 0, -128.0, 127.0, 0, 0, 0, 0)
        scale = torch.full((1,), 3.0 / 128.0, device=dev)
        zp = torch.zeros(1, device=dev)
        t0 = timed(lambda: nat.stats(nat.STAT_ABSMAX, x, 1, 1, n))
        t1 = timed(lambda: nat.fakequant_fwd(d, x, scale, zp))
        t2 = timed(lambda: nat.fakequant_bwd(d, g, x, scale, zp, True, False))
        t3 = timed(lambda: nat.fakequant_bwd(d, g, x, scale, zp, False, False))
        print('%-4s per-tensor: absmax %.4f ms %.2f TB/s | fwd %.4f ms %.2f TB/s | bwd(dscale) %.4f ms %.2f TB/s | '
              'bwd(dx) %.4f ms %.2f TB/s' % (name, t0, b * n / t0 / 1e9, t1, 2 * b * n / t1 / 1e9, t2, 3 * b * n / t2 / 1e9,
                                            t3, 3 * b * n / t3 / 1e9), flush=True)
        del x, g


if __name__ == '__main__':
    main()
