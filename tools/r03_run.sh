out=gpurun_out/r03q
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_moments.py tests/test_gpu_onepass.py -q -m gpu -x > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r03q/cols_moments.txt
import sys, torch
sys.path.insert(0, '.')
from brevitas_amd import _native as nat
def timeit(fn, iters=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (outer, ch, inner), dt in (((802816, 512, 1), torch.bfloat16), ((65536, 4096, 1), torch.bfloat16), ((65536, 4096, 1), torch.float32), ((1024, 2048, 49), torch.bfloat16)):
    n = outer * ch * inner
    x = torch.randn(n, device='cuda', dtype=dt)
    ca, cb = torch.randn(ch, device='cuda'), torch.randn(ch, device='cuda')
    b = x.element_size()
    t0 = timeit(lambda: nat.stats(nat.STAT_ABSMAX, x, outer, ch, inner))
    t1 = timeit(lambda: nat.abs_moments(x, outer, ch, inner))
    t2 = timeit(lambda: nat.abs_affine_bwd(x, ca, cb, outer, ch, inner))
    print('[%d,%d,%d] %s: abs-max %.3f ms %.2f TB/s | moments %.3f ms %.2f TB/s (%.2f x abs-max) | moments backward %.3f ms %.2f TB/s' % (
        outer, ch, inner, str(dt).replace('torch.', ''), t0, n * b / t0 / 1e9, t1, n * b / t1 / 1e9, t1 / t0, t2, 2 * n * b / t2 / 1e9))
PY
