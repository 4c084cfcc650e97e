"""Experiment: do the three streams of the backward (read g, read x, write dx) collide in HBM channels /
banks when the tensors share their alignment?  The caching allocator hands out 2 MiB-aligned blocks, so
element i of x, g and dx has the same low address bits.  This script times bvq_fakequant_bwd with g and dx
shifted by a few KB inside over-allocated buffers (still 16-byte aligned).  Developer tool."""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    N, C, HW = 256, 512, 56 * 56
    n = N * C * HW
    dt = torch.bfloat16
    code = nat.dtype_code(dt)
    lib = nat.lib
    stream = nat.stream_ptr(dev)
    pad = 1 << 22
    x = torch.randn(n, device=dev, dtype=dt)
    gbuf = torch.randn(n + pad, device=dev, dtype=dt)
    dbuf = torch.empty(n + pad, device=dev, dtype=dt)
    stat = nat.stats(nat.STAT_ABSMAX, x, N, C, HW)
    scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
    zp = torch.zeros(1, device=dev)
    d = nat.QuantDesc(N, C, HW, code, code, code, 0, 1, 0, -128.0, 127.0, 0, 0, 0, 0)
    ds = torch.empty(C, device=dev, dtype=torch.float32)
    wsb = int(lib.bvq_fakequant_bwd_workspace_bytes(ctypes.byref(d)))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for name, go, do in (('aligned', 0, 0), ('g+256B', 128, 0), ('dx+256B', 0, 128), ('g+4K dx+8K', 2048, 4096),
                         ('g+64K+256 dx+128K+512', 32768 + 128, 65536 + 256), ('g+1M dx+2M+4K', 1 << 19, (1 << 20) + 2048),
                         ('aligned again', 0, 0)):
        g = gbuf[go:go + n]
        dx = dbuf[do:do + n]

        def run():
            nat.check(lib.bvq_fakequant_bwd(ctypes.byref(d), nat.ptr(g), nat.ptr(x), nat.ptr(scale), nat.ptr(zp),
                                            nat.ptr(dx), nat.ptr(ds), None, None, None, nat.ptr(ws), wsb, stream), 'bwd')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(10):
            run()
        en.record()
        torch.cuda.synchronize()
        t = st.elapsed_time(en) / 10
        print('%-28s bwd %.3f ms  %.2f TB/s' % (name, t, 3 * 2 * n / t / 1e9), flush=True)


if __name__ == '__main__':
    main()
