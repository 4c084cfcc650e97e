"""Developer tool: the chip's own ceiling for the three access mixes of the headline step, next to the product kernels.

    python tools/yardstick.py build      # here (no GPU): hipcc tools/yardstick.hip -> build/tools/libyardstick.so
    python tools/yardstick.py            # on the GPU box

tools/yardstick.hip moves the headline tensor's bytes with NO arithmetic: one read stream (what the abs-max does),
one read + one write (the forward), two reads + one write (the backward); 16 bytes per lane, all of a wave's loads
issued first, non-temporal or default policy, short-lived waves (the product's decomposition) or a persistent grid.
For every mix the best of the sweep is the empirical ceiling; the product kernels are timed in the same process, in
interleaved rounds, on the same buffers.  HIP events on the launching stream, median over the rounds."""
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'build', 'tools', 'libyardstick.so')


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           os.path.join(ROOT, 'tools', 'yardstick.hip'), '-o', SO]
    subprocess.check_call(cmd)
    print('built', SO)


def main():
    import torch
    sys.path.insert(0, ROOT)
    from brevitas_amd import _native as nat
    lib = ctypes.CDLL(SO)
    lib.yardstick.restype = ctypes.c_int
    lib.yardstick.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_void_p]
    dev = 'cuda:0'
    N, C, H, W = 256, 512, 56, 56
    dt = torch.bfloat16
    x = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    g = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
    o = torch.empty_like(x)
    sink = torch.zeros(4, device=dev, dtype=torch.int32)
    nbytes = x.numel() * 2
    stream = torch.cuda.current_stream().cuda_stream

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def timed(fn):
        a = ev()
        fn()
        b = ev()
        return a, b

    def yard(mode, nt, ch, form, blocks):
        rc = lib.yardstick(mode, nt, ch, form, blocks, x.data_ptr(), g.data_ptr(), o.data_ptr(), sink.data_ptr(),
                           nbytes, stream)
        assert rc == 0, rc

    # correctness of the yardstick itself (it must really move the bytes)
    yard(2, 1, 8, 0, 0)
    want = (x.view(torch.int16) ^ g.view(torch.int16))
    assert torch.equal(o.view(torch.int16), want), 'triad kernel wrong'
    yard(1, 1, 4, 1, 2048)
    assert torch.equal(o.view(torch.int16), x.view(torch.int16)), 'copy kernel wrong'

    code = nat.dtype_code(dt)
    d = nat.QuantDesc(N, C, H * W, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    zp = torch.zeros(1, device=dev)
    stat = nat.stats(nat.STAT_ABSMAX, x, N, C, H * W)
    scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)

    cands = {}
    for mode, mname in ((0, 'read'), (1, 'copy'), (2, 'triad')):
        for nt in (1, 0):
            for ch in (2, 4, 8):
                cands['%s unit nt=%d ch=%d' % (mname, nt, ch)] = (lambda m=mode, n=nt, c=ch: yard(m, n, c, 0, 0))
                for blocks in (1024, 2048, 4096):
                    cands['%s persistent nt=%d ch=%d blocks=%d' % (mname, nt, ch, blocks)] = \
                        (lambda m=mode, n=nt, c=ch, bl=blocks: yard(m, n, c, 1, bl))
    cands['product abs-max (bvq_stats, kernel + finish)'] = lambda: nat.stats(nat.STAT_ABSMAX, x, N, C, H * W)
    cands['product forward (bvq_fakequant_fwd)'] = lambda: nat.fakequant_fwd(d, x, scale, zp)
    cands['product backward (bvq_fakequant_bwd, kernel + finish)'] = \
        lambda: nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, scale.dtype, 128.0, scale.dtype)
    cands['torch copy_'] = lambda: o.copy_(x)
    cands['torch add (2R1W)'] = lambda: torch.add(x, g, out=o)

    for fn in cands.values():       # warm-up, clock settling
        fn()
    for _ in range(30):
        yard(2, 1, 8, 0, 0)
    res = {k: [] for k in cands}
    for rnd in range(7):
        pairs = []
        for k, fn in cands.items():
            fn()
            pairs.append((k, timed(fn)))
        torch.cuda.synchronize()
        for k, (a, b) in pairs:
            res[k].append(a.elapsed_time(b))
    passes = {'read': 1, 'copy': 2, 'triad': 3, 'product abs-max': 1, 'product forward': 2, 'product backward': 3,
              'torch copy_': 2, 'torch add': 3}
    print('# tools/yardstick.py: [256,512,56,56] bf16 (822 MB per stream), one MI355X; median / min ms over 7 interleaved rounds')
    best = {}
    for k, ts in res.items():
        p = next(v for n, v in passes.items() if k.startswith(n))
        med, mn = statistics.median(ts), min(ts)
        tb = p * nbytes / med / 1e9
        print('%-58s %.4f / %.4f ms  %5.2f TB/s' % (k, med, mn, tb))
        kind = k.split(' ')[0]
        if kind in ('read', 'copy', 'triad') and (kind not in best or med < best[kind][1]):
            best[kind] = (k, med, tb)
    print()
    for kind, prod in (('read', 'product abs-max (bvq_stats, kernel + finish)'), ('copy', 'product forward (bvq_fakequant_fwd)'),
                       ('triad', 'product backward (bvq_fakequant_bwd, kernel + finish)')):
        k, med, tb = best[kind]
        pm = statistics.median(res[prod])
        print('ceiling %-5s: %.4f ms (%.2f TB/s, %s)  |  %s: %.4f ms = %.3f x the ceiling'
              % (kind, med, tb, k, prod.split(' (')[0], pm, pm / med))
    tot_c = sum(best[k][1] for k in ('read', 'copy', 'triad'))
    print('sum of the three ceilings: %.4f ms per step = %.1f Gelem/s = %.3f of the 12 B/elem roofline at 8 TB/s'
          % (tot_c, x.numel() / tot_c / 1e6, 12 * x.numel() / (tot_c * 1e-3) / 8e12))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        main()
