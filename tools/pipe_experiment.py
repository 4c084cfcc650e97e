"""Slab-pipelined statistic + quantizer (bvq_stats_fakequant_fwd, large per-channel tensors) against the two-kernel
route on the headline activation: time per forward over a sweep of the pipeline's knobs (environment variables read
per call by libbvq: BVQ_PIPE_SLAB_KB, BVQ_PIPE_LAG_KB, BVQ_PIPE_RPU_S and, in a -DBVQ_PIPE_EXPERIMENT build, the
cache policies BVQ_PIPE_SNT / QNTL / QNTS).  Developer tool, not the judged bench.

    python tools/pipe_experiment.py [quick]
"""
import ctypes
import itertools
import os
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def main():
    quick = 'quick' in sys.argv
    dev = torch.device('cuda', 0)
    N, C, HW = 256, 512, 56 * 56
    dt = torch.bfloat16
    code = nat.dtype_code(dt)
    lib = nat.lib
    stream = nat.stream_ptr(dev)
    torch.manual_seed(123456)
    x = torch.randn(N, C, HW, device=dev, dtype=dt)
    y0 = torch.empty_like(x)
    y1 = torch.empty_like(x)
    zp = torch.zeros(1, device=dev)
    stat0 = torch.empty(C, device=dev, dtype=dt)
    scale0 = torch.empty(C, device=dev, dtype=dt)
    stat1 = torch.empty(C, device=dev, dtype=dt)
    scale1 = torch.empty(C, device=dev, dtype=dt)
    wsb = int(lib.bvq_stats_workspace_bytes(nat.STAT_ABSMAX, code, N, C, HW))
    ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
    d = nat.QuantDesc(N, C, HW, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    pws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)

    def two():
        nat.check(lib.bvq_absmax_scale(nat.PRE_NONE, code, nat.ptr(x), N, C, HW, nat.ptr(stat0), 1e-10, 1, 128.0, code,
                                       nat.ptr(scale0), nat.ptr(ws), ws.numel(), stream), 'absmax')
        nat.check(lib.bvq_fakequant_fwd(ctypes.byref(d), nat.ptr(x), nat.ptr(scale0), nat.ptr(zp), nat.ptr(y0), None,
                                        stream), 'fwd')

    def pipe():
        nat.check(lib.bvq_stats_fakequant_fwd(ctypes.byref(d), nat.ptr(x), 1e-10, 1, 128.0, nat.ptr(stat1),
                                              nat.ptr(scale1), nat.ptr(y1), nat.ptr(pws), pws.numel(), stream), 'pipe')

    def timeit(fn, iters=20):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        return st.elapsed_time(en) / iters

    for _ in range(30):  # clocks
        two()
    t2 = timeit(two)
    print('two-kernel route                         %.4f ms  (%.2f TB/s algorithmic, 3 passes)' % (
        t2, 3 * x.numel() * 2 / t2 / 1e9), flush=True)
    slabs = [12288] if quick else [3200, 6400, 12800, 25600, 51200]
    lags = [98304] if quick else [32768, 65536, 98304, 131072, 163840]
    rpus = [0] if quick else [1, 2, 4, 8]
    pols = [(0, 1, 1)] if quick else [(0, 1, 1), (0, 0, 1), (1, 1, 1), (0, 1, 0), (0, 0, 0)]
    results = []

    def run(slab, lag, rpu, pol, check=False, lds=0):
        os.environ['BVQ_PIPE_LDS_KB'] = str(lds)
        os.environ['BVQ_PIPE_SLAB_KB'] = str(slab)
        os.environ['BVQ_PIPE_LAG_KB'] = str(lag)
        os.environ['BVQ_PIPE_RPU_S'] = str(rpu)
        os.environ['BVQ_PIPE_SNT'], os.environ['BVQ_PIPE_QNTL'], os.environ['BVQ_PIPE_QNTS'] = map(str, pol)
        if int(lib.bvq_stats_fakequant_fwd_workspace_bytes(ctypes.byref(d), nat.ptr(x), nat.ptr(y1))) <= 16:
            return None
        if check:
            y1.zero_()
            pipe()
            torch.cuda.synchronize()
            ok = torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(stat0, stat1) and \
                torch.equal(scale0.view(torch.int16), scale1.view(torch.int16))
            if not ok:
                print('MISMATCH slab=%d lag=%d rpu=%d pol=%s' % (slab, lag, rpu, pol), flush=True)
        t = timeit(pipe)
        results.append((t, slab, lag, rpu, pol, lds))
        print('pipe slab=%6d KB lag=%6d KB rpu_s=%d snt/qntl/qnts=%s lds=%2d KB  %.4f ms  (%.2fx)' % (
            slab, lag, rpu, pol, lds, t, t2 / t), flush=True)
        return t

    if 'occ' in sys.argv:
        # stage 2: cap the resident waves (unused LDS) so that a short lag suffices
        for lds in (64, 53, 40, 0):
            for slab in (6400, 12800):
                for rpu in (2, 4):
                    for lag in (16384, 24576, 32768, 49152, 65536, 98304):
                        run(slab, lag, rpu, (0, 1, 1), check=True, lds=lds)
        print('two-kernel route again                   %.4f ms' % timeit(two), flush=True)
        print('best: %.4f ms slab=%d lag=%d rpu=%d pol=%s lds=%d' % min(results))
        return

    # stage 1: slab x lag at default rows and policy
    for slab, lag in itertools.product(slabs, lags):
        run(slab, lag, 0, (0, 1, 1), check=True)
    if not quick and results:
        best = min(results)
        _, slab, lag, _, _, _ = best
        for rpu in rpus:
            run(slab, lag, rpu, (0, 1, 1), check=True)
        best = min(results)
        _, slab, lag, rpu, _, _ = best
        for pol in pols[1:]:
            run(slab, lag, rpu, pol, check=True)
    t2b = timeit(two)
    print('two-kernel route again                   %.4f ms' % t2b, flush=True)
    if results:
        print('best: %.4f ms slab=%d lag=%d rpu=%d pol=%s lds=%d' % min(results))


if __name__ == '__main__':
    main()
