import os, sys, torch
sys.path.insert(0, '.')
from brevitas_amd import _native as nat
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
n = 256 * 512 * 56 * 56
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(n, device='cuda:0', dtype=dt)
    k1, k2 = int(1e-5 * n) + 1, int(0.99999 * n + 0.5)
    t2 = timeit(lambda: (nat.kth_value(x, k1, 1, 1, n, False), nat.kth_value(x, k2, 1, 1, n, False)))
    t1 = timeit(lambda: nat.kth_pair(x, k1, k2, 1, 1, n, False))
    print('%s PercentileInterval ranks of [256,512,56,56]: two selections %.3f ms, bvq_kth_pair %.3f ms' % (str(dt)[6:], t2, t1), flush=True)
