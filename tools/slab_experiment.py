"""Experiment: does the 256 MiB Infinity Cache (MALL) pay for re-reading x between the statistic and
the quantizer if the two kernels run back to back on a channel slab that fits in it?

The per-channel statistic of channel c depends on channel c alone, so a [N,C,H,W] activation can be
processed slab by slab (statistic then quantize) instead of statistic-over-everything then
quantize-over-everything.  This script compares the two schedules on separately allocated slabs
[N, C/S, H*W] (no kernel changes needed), S in {1, 4, 8, 16}, with raw C-ABI calls and
preallocated outputs so that host overhead does not blur it.  Developer tool, not the judged bench."""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    N, C, HW = 256, 512, 56 * 56
    dt = torch.bfloat16
    code = nat.dtype_code(dt)
    lib = nat.lib
    stream = nat.stream_ptr(dev)
    zp = torch.zeros(1, device=dev)
    for S in (1, 4, 8, 16):
        cs = C // S
        xs = [torch.randn(N, cs, HW, device=dev, dtype=dt) for _ in range(S)]
        ys = [torch.empty_like(x) for x in xs]
        stats = [torch.empty(cs, device=dev, dtype=dt) for _ in range(S)]
        scales = [torch.empty(cs, device=dev, dtype=dt) for _ in range(S)]
        wsb = int(lib.bvq_stats_workspace_bytes(nat.STAT_ABSMAX, code, N, cs, HW))
        wss = [torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev) for _ in range(S)]
        d = nat.QuantDesc(N, cs, HW, code, code, code, 0, 1, 0, -128.0, 127.0, 0, 0, 0, 0)

        def stat(i):
            nat.check(lib.bvq_absmax_scale(nat.PRE_NONE, code, nat.ptr(xs[i]), N, cs, HW, nat.ptr(stats[i]), 1e-10, 1,
                                           128.0, code, nat.ptr(scales[i]), nat.ptr(wss[i]), wss[i].numel(), stream),
                      'absmax')

        def fwd(i):
            nat.check(lib.bvq_fakequant_fwd(ctypes.byref(d), nat.ptr(xs[i]), nat.ptr(scales[i]), nat.ptr(zp),
                                            nat.ptr(ys[i]), None, stream), 'fwd')

        def interleaved():
            for i in range(S):
                stat(i)
                fwd(i)

        def phased():
            for i in range(S):
                stat(i)
            for i in range(S):
                fwd(i)

        for name, fn in (('phased     ', phased), ('interleaved', interleaved)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(10):
                fn()
            en.record()
            torch.cuda.synchronize()
            t = st.elapsed_time(en) / 10
            print('slabs=%2d (%5.1f MB each) %s stat+fwd %.3f ms  (%.2f TB/s algorithmic)' % (
                S, xs[0].numel() * 2 / 1e6, name, t, 3 * 2 * N * C * HW / t / 1e9), flush=True)
        del xs, ys


if __name__ == '__main__':
    main()
