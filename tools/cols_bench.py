"""Column-mapped kernels (channel axis last or nearly last) through the raw C ABI: abs-max, forward, and the
stats-scaled backward (bvq_fakequant_bwd_stats), one line per shape.  BREVITAS_AMD_LIB selects a variant build."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brevitas_amd import _native as nat  # noqa: E402


def timeit(fn, iters=30, warm=8):
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


SHAPES = [((200704, 512, 1), torch.bfloat16), ((802816, 512, 1), torch.bfloat16), ((1024, 2048, 49), torch.bfloat16),
          ((65536, 4096, 1), torch.bfloat16), ((65536, 4096, 1), torch.float16), ((65536, 4096, 1), torch.float32)]


def main():
    global SHAPES
    # python tools/cols_bench.py 1024,1024,196,bf16 2048,2048,49,bf16 ...: other [outer, channels, inner] layouts
    picked = [a for a in sys.argv[1:] if a.count(',') == 3]
    if picked:
        names = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}
        SHAPES = [(tuple(int(v) for v in a.split(',')[:3]), names[a.split(',')[3]]) for a in picked]
    dev = 'cuda:0'
    print(os.environ.get('BREVITAS_AMD_LIB', 'libbvq.so (default build)'))
    for (outer, ch, inner), dt in SHAPES:
        n = outer * ch * inner
        b = torch.empty(0, dtype=dt).element_size()
        x = torch.randn(n, device=dev, dtype=dt)
        g = torch.randn(n, device=dev, dtype=dt)
        t_s = timeit(lambda: nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner))
        stat = nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner)
        scale = (stat.float().clamp_min(1e-10) / 128.0).to(dt)
        zp = torch.zeros(1, device=dev)
        d = nat.QuantDesc(outer, ch, inner, nat.dtype_code(dt), nat.dtype_code(dt), nat.dtype_code(dt), 0, 1, 0,
                          -128.0, 127.0, 0, 0, 0, 0)
        t_f = timeit(lambda: nat.fakequant_fwd(d, x, scale, zp))
        assert nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt) is not None
        t_b = timeit(lambda: nat.fakequant_bwd_stats(d, g, x, scale, zp, stat, dt, 128.0, dt))
        t_p = timeit(lambda: nat.fakequant_bwd(d, g, x, scale, zp, True, False))
        print('[%d,%d,%d] %s: absmax %.3f ms %.2f TB/s | fwd %.3f ms %.2f TB/s | bwd(stats) %.3f ms %.2f TB/s | '
              'bwd(dscale) %.3f ms %.2f TB/s' % (outer, ch, inner, str(dt)[6:], t_s, b * n / t_s / 1e9, t_f,
                                                 2 * b * n / t_f / 1e9, t_b, 3 * b * n / t_b / 1e9, t_p,
                                                 3 * b * n / t_p / 1e9), flush=True)
        del x, g


if __name__ == '__main__':
    main()
