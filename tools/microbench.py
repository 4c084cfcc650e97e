"""Developer microbenchmark of the raw C-ABI kernels (not the judged bench; see bench.py)."""
import sys
import time

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def main():
    dev = 'cuda:0'
    N, C, H, W = 256, 512, 56, 56
    n = N * C * H * W
    print(torch.cuda.get_device_name(0), 'elements', n)
    for dt, name in ((torch.bfloat16, 'bf16'), (torch.float16, 'f16'), (torch.float32, 'f32')):
        x = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
        g = torch.randn(N, C, H, W, device=dev, dtype=dt).reshape(-1)
        b = x.element_size()
        # device copy as the achievable-bandwidth yardstick
        y = torch.empty_like(x)
        t = timeit(lambda: y.copy_(x))
        print('%s copy           %.3f ms  %.2f TB/s' % (name, t, 2 * b * n / t / 1e9))
        t = timeit(lambda: nat.unary(nat.OP_ABS, x))
        print('%s |x| map kernel  %.3f ms  %.2f TB/s (grid-stride, default cache policy)' % (name, t, 2 * b * n / t / 1e9))
        k = int(0.99999 * n + 0.5)
        t = timeit(lambda: nat.kth_value(x, k, 1, 1, n, True))
        passes = 2 if dt == torch.float32 else 1  # per-tensor route: 15-bit LDS digit (+ one read for the low bits)
        print('%s percentile(99.999) of |x| per-tensor %.3f ms  %.2f TB/s (%d reads of x; torch.kthvalue: see below)' % (
            name, t, passes * b * n / t / 1e9, passes))
        kl = int(0.001 * n) + 1
        t = timeit(lambda: nat.kth_value(x, kl, 1, 1, n, False))
        print('%s percentile(0.1) of x (signed keys) per-tensor %.3f ms' % (name, t))
        if dt == torch.bfloat16:
            t0 = timeit(lambda: x.abs().kthvalue(k), iters=2, warm=1)
            print('%s torch: x.abs().kthvalue(k)          %.3f ms' % (name, t0))
        for tag, ch in (('per-tensor', 1), ('per-channel', C)):
            outer, inner = (N, H * W) if ch > 1 else (1, n)
            t = timeit(lambda: nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner))
            print('%s absmax %-11s %.3f ms  %.2f TB/s' % (name, tag, t, b * n / t / 1e9))
            stat = nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner)
            scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
            zp = torch.zeros(1, device=dev)
            d = nat.QuantDesc(outer, ch, inner, nat.dtype_code(dt), nat.dtype_code(dt), nat.dtype_code(dt), 0,
                              int(ch > 1), 0, -128.0, 127.0, 0, 0, 0, 0)
            if nat.stats_fakequant_fwd(d, x, 1e-10, 128.0, dt) is not None:
                t = timeit(lambda: nat.stats_fakequant_fwd(d, x, 1e-10, 128.0, dt))
                print('%s stat+fwd %-9s %.3f ms  %.2f TB/s (one kernel: x read once, 2 passes)' % (
                    name, tag, t, 2 * b * n / t / 1e9))
            t = timeit(lambda: nat.fakequant_fwd(d, x, scale, zp))
            print('%s fwd    %-11s %.3f ms  %.2f TB/s' % (name, tag, t, 2 * b * n / t / 1e9))
            t = timeit(lambda: nat.fakequant_bwd(d, g, x, scale, zp, True, False))
            print('%s bwd    %-11s %.3f ms  %.2f TB/s (with dscale)' % (name, tag, t, 3 * b * n / t / 1e9))
            t = timeit(lambda: nat.fakequant_bwd(d, g, x, scale, zp, True, False, tie_stat=stat))
            print('%s bwd    %-11s %.3f ms  %.2f TB/s (with dscale + arg-max search: the stats-scaled graph)' % (
                name, tag, t, 3 * b * n / t / 1e9))
            t = timeit(lambda: nat.fakequant_bwd(d, g, x, scale, zp, False, False))
            print('%s bwd    %-11s %.3f ms  %.2f TB/s (dx only)' % (name, tag, t, 3 * b * n / t / 1e9))
        del x, g, y


if __name__ == '__main__' and 'small' not in sys.argv:
    main()


def small_tensor_latency():
    """weights are launch/latency-bound: wall-clock per fwd+bwd of the module-level quantizers"""
    import time

    import brevitas_amd.quant as Q
    dev = 'cuda:0'
    for name, shape, dt, bw in (('conv W [512,512,3,3] f32 int8', (512, 512, 3, 3), torch.float32, 8),
                                ('linear W [8192,8192] bf16 int4', (8192, 8192), torch.bfloat16, 4),
                                ('conv W [256,1024,1,1] bf16 int8', (256, 1024, 1, 1), torch.bfloat16, 8)):
        w = torch.nn.Parameter((torch.randn(shape, device=dev) * 0.02).to(dt))
        g = torch.randn(shape, device=dev).to(dt)
        q = Q.Int8WeightPerChannelFloat(w, bw).to(dev)

        def step():
            w.grad = None
            y = q(w)[0]
            y.backward(g)

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e6
        t_gpu = timeit(step, iters=20, warm=2) * 1e3
        print('%-34s wall %.0f us / step, device %.0f us / step (%d elements)' % (name, wall, t_gpu, w.numel()))
        # the same step captured once in a HIP graph and replayed: launch-bound work without the host
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        w.grad = None
        with torch.cuda.graph(graph):
            y = q(w)[0]
            y.backward(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        wall_g = (time.perf_counter() - t0) / n * 1e6
        print('%-34s HIP graph replay: wall %.0f us / step' % ('', wall_g))


if __name__ == '__main__' and 'small' in sys.argv:
    small_tensor_latency()
