"""Developer A/B harness: interleaved rounds of the headline kernel sequence over several builds of
libbvq.so (build-time experiment flags, brevitas_amd/csrc/build.py -D... --out=...), in ONE process.

    python tools/variant_bench.py build/variants/libbvq_*.so [cases=pc_bf16,pt_bf16,pc_f16,pc_f32]

Cases: pc_* = per-channel [256,512,56,56] stats-scaled graph (statistic, forward, backward with the arg-max search);
pt_bf16 = per-tensor learned-scale steady state (forward, backward with dscale only)."""
import glob
import statistics
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402

DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('cases=')]
    cases = [a.split('=', 1)[1].split(',') for a in sys.argv[1:] if a.startswith('cases=')]
    cases = cases[0] if cases else ['pc_bf16']
    paths = [nat.LIB_PATH] + [p for a in args for p in sorted(glob.glob(a))]
    libs = [(p.split('libbvq')[-1].replace('.so', '').strip('_') or 'base', nat._load(p, strict=False)) for p in paths]
    dev = 'cuda:0'
    N, C, H, W = 256, 512, 56, 56
    zp = torch.zeros(1, device=dev)

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for case in cases:
        layout, dn = case.split('_')
        dt = DT[dn]
        code = nat.dtype_code(dt)
        x = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
        g = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
        pc = layout == 'pc'
        d = nat.QuantDesc(N, C, H * W, code, code, code if pc else nat.F32, nat.F32, int(pc), 0, -128.0, 127.0, 0, 0, 0, 0)
        res = {name: {'absmax': [], 'fwd': [], 'bwd': [], 'step': []} for name, _ in libs}
        for rnd in range(8):
            for name, lib in libs:
                nat.lib = lib
                for it in range(3):
                    e0 = ev()
                    if pc:
                        stat = nat.stats(nat.STAT_ABSMAX, x, N, C, H * W)
                        e1 = ev()
                        scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
                    else:
                        e1 = ev()
                        scale = torch.full((1,), 3.0 / 128.0, device=dev)
                    e2 = ev()
                    y = nat.fakequant_fwd(d, x, scale, zp)
                    e3 = ev()
                    if pc:
                        out = nat.fakequant_bwd(d, g, x, scale, zp, True, False, tie_stat=stat)
                    else:
                        out = nat.fakequant_bwd(d, g, x, scale, zp, True, False)
                    e4 = ev()
                    torch.cuda.synchronize()
                    del y, out
                    if it == 0:
                        continue  # warm-up of this variant in this round
                    r = res[name]
                    r['absmax'].append(e0.elapsed_time(e1))
                    r['fwd'].append(e2.elapsed_time(e3))
                    r['bwd'].append(e3.elapsed_time(e4))
                    r['step'].append(e0.elapsed_time(e4))
        print('== %s' % case)
        print('%-10s %18s %18s %18s %18s' % ('variant', 'absmax med/min', 'fwd med/min', 'bwd med/min', 'seq med/min'))
        for name, _ in libs:
            r = res[name]
            print('%-10s ' % name + ' '.join('%8.3f /%8.3f' % (statistics.median(r[k]), min(r[k]))
                                             for k in ('absmax', 'fwd', 'bwd', 'step')), flush=True)
        del x, g


if __name__ == '__main__':
    main()
