"""Developer A/B harness: interleaved rounds of the headline kernel sequence over several builds of
libbvq.so (build-time experiment flags, brevitas_amd/csrc/build.py -D... --out=...), in ONE process.

    python tools/variant_bench.py build/variants/libbvq_*.so
"""
import glob
import statistics
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def main():
    paths = [nat.LIB_PATH] + [p for a in sys.argv[1:] for p in sorted(glob.glob(a))]
    libs = [(p.split('libbvq')[-1].replace('.so', '').strip('_') or 'base', nat._load(p)) for p in paths]
    dev = 'cuda:0'
    N, C, H, W = 256, 512, 56, 56
    n = N * C * H * W
    dt = torch.bfloat16
    x = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    g = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    zp = torch.zeros(1, device=dev)
    d = nat.QuantDesc(N, C, H * W, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    res = {name: {'absmax': [], 'fwd': [], 'bwd': [], 'step': []} for name, _ in libs}

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for rnd in range(8):
        for name, lib in libs:
            nat.lib = lib
            for it in range(3):
                e0 = ev()
                stat = nat.stats(nat.STAT_ABSMAX, x, N, C, H * W)
                e1 = ev()
                scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
                e2 = ev()
                y = nat.fakequant_fwd(d, x, scale, zp)
                e3 = ev()
                dx, ds, _, info = nat.fakequant_bwd(d, g, x, scale, zp, True, False, tie_stat=stat)
                e4 = ev()
                torch.cuda.synchronize()
                if it == 0:
                    continue  # warm-up of this variant in this round
                r = res[name]
                r['absmax'].append(e0.elapsed_time(e1))
                r['fwd'].append(e2.elapsed_time(e3))
                r['bwd'].append(e3.elapsed_time(e4))
                r['step'].append(e0.elapsed_time(e4))
                del y, dx
    print('%-10s %18s %18s %18s %18s' % ('variant', 'absmax med/min', 'fwd med/min', 'bwd med/min', 'seq med/min'))
    for name, _ in libs:
        r = res[name]
        print('%-10s ' % name + ' '.join('%8.3f /%8.3f' % (statistics.median(r[k]), min(r[k]))
                                         for k in ('absmax', 'fwd', 'bwd', 'step')))


if __name__ == '__main__':
    main()
