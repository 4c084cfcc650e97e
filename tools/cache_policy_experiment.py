"""Where a pass over the activation starts, and with which cache policy it loads x: can the next pass find the
previous pass's last ~200 MB of x in the 256 MiB Infinity Cache?  Needs a -DBVQ_CACHE_EXPERIMENT build of libbvq.so
(knobs read from the environment per call: BVQ_X_{STAT,FWD,BWD}_REV walk a pass from the tensor's end,
BVQ_X_STAT_NT / BVQ_X_FWD_NTL / BVQ_X_BWD_NTX choose non-temporal (1) or allocating (0) loads of x, BVQ_X_FWD_NTS /
BVQ_X_BWD_NT the policy of the other streams).  Times the statistic -> forward -> backward sequence of the headline
step back to back (steady state, like bench.py's loop), interleaving the configurations over several rounds.

    python -m brevitas_amd.csrc.build -DBVQ_CACHE_EXPERIMENT --out=build/variants/libbvq_cache.so
    python tools/cache_policy_experiment.py build/variants/libbvq_cache.so
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402

KNOBS = ('STAT_REV', 'FWD_REV', 'BWD_REV', 'STAT_NT', 'FWD_NTL', 'FWD_NTS', 'BWD_NTX', 'BWD_NT')
BASE = dict(STAT_REV=0, FWD_REV=0, BWD_REV=0, STAT_NT=1, FWD_NTL=1, FWD_NTS=1, BWD_NTX=1, BWD_NT=1)
CONFIGS = [
    ('base: all forward, all nt', {}),
    ('zigzag s< f> b<, all nt', dict(STAT_REV=1, BWD_REV=1)),
    ('zigzag s< f> b<, x allocating', dict(STAT_REV=1, BWD_REV=1, STAT_NT=0, FWD_NTL=0, BWD_NTX=0)),
    ('zigzag s> f< b>, x allocating', dict(FWD_REV=1, STAT_NT=0, FWD_NTL=0, BWD_NTX=0)),
    ('zigzag s> f< b>, stat+fwd x allocating', dict(FWD_REV=1, STAT_NT=0, FWD_NTL=0)),
    ('f> b< only, fwd x allocating', dict(BWD_REV=1, FWD_NTL=0)),
    ('s> f< only, stat allocating', dict(FWD_REV=1, STAT_NT=0)),
    ('zigzag s< f> b<, everything allocating', dict(STAT_REV=1, BWD_REV=1, STAT_NT=0, FWD_NTL=0, FWD_NTS=0, BWD_NTX=0,
                                                    BWD_NT=0)),
    ('all forward, x allocating', dict(STAT_NT=0, FWD_NTL=0, BWD_NTX=0)),
]


def main():
    nat.lib = nat._load(os.path.abspath(sys.argv[1]))
    dev = 'cuda:0'
    N, C, H, W = 256, 512, 56, 56
    dt = torch.bfloat16
    x = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    g = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    zp = torch.zeros(1, device=dev)
    d = nat.QuantDesc(N, C, H * W, nat.BF16, nat.BF16, nat.BF16, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    res = {name: {'stat': [], 'fwd': [], 'bwd': [], 'step': []} for name, _ in CONFIGS}
    ref = None

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def step():
        e0 = ev()
        stat, scale = nat.absmax_scale(x, N, C, H * W, 1e-10, 128.0, dt)
        e1 = ev()
        y = nat.fakequant_fwd(d, x, scale, zp)
        e2 = ev()
        dx = nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt)
        e3 = ev()
        return (e0, e1, e2, e3), y, dx

    for rnd in range(6):
        for name, over in CONFIGS:
            for k in KNOBS:
                os.environ['BVQ_X_' + k] = str(dict(BASE, **over)[k])
            evs = []
            for it in range(6):   # back-to-back steps: the cache state of step i is what step i - 1 left
                e, y, dx = step()
                evs.append(e)
                if it < 5:
                    del y, dx
            torch.cuda.synchronize()
            if ref is None:
                ref = (y.clone(), dx.clone())
            else:
                assert torch.equal(y.view(torch.int16), ref[0].view(torch.int16)), name
                assert torch.equal(dx.view(torch.int16), ref[1].view(torch.int16)), name
            del y, dx
            r = res[name]
            for e0, e1, e2, e3 in evs[2:]:
                r['stat'].append(e0.elapsed_time(e1))
                r['fwd'].append(e1.elapsed_time(e2))
                r['bwd'].append(e2.elapsed_time(e3))
            r['step'].append(evs[2][0].elapsed_time(evs[-1][3]) / 4)
    print('%-46s %16s %16s %16s %16s' % ('configuration', 'stat med/min', 'fwd med/min', 'bwd med/min', 'step med/min'))
    for name, _ in CONFIGS:
        r = res[name]
        print('%-46s ' % name + ' '.join('%7.3f /%7.3f' % (statistics.median(r[k]), min(r[k]))
                                         for k in ('stat', 'fwd', 'bwd', 'step')), flush=True)


if __name__ == '__main__':
    main()
