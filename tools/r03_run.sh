out=gpurun_out/r03w
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_channel_last_golden.py tests/test_percentile_golden.py tests/test_gpu_select_onepass.py tests/test_gpu_distributed.py tests/test_gpu_ste_stats_modules.py -q -m gpu -x > $out/tests.txt 2>&1 || { tail -40 $out/tests.txt; exit 1; }
tail -3 $out/tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r03w/cols_percentile.txt
import sys, torch
sys.path.insert(0, '.')
from brevitas_amd import _native as nat
from brevitas_amd.core.stats import AbsPercentile
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (outer, ch), dt in (((802816, 512), torch.bfloat16), ((65536, 4096), torch.bfloat16), ((65536, 4096), torch.float32)):
    x = torch.randn(outer, ch, device='cuda', dtype=dt)
    k = int(0.99999 * outer + 0.5)
    t_abs = timeit(lambda: nat.stats(nat.STAT_ABSMAX, x.reshape(-1), outer, ch, 1))
    t_sel = timeit(lambda: nat.kth_value(x.reshape(-1), k, outer, ch, 1, True))
    xi = x.clone().requires_grad_(True)
    m = AbsPercentile(99.999, 0)
    def fb():
        xi.grad = None
        m(xi).sum().backward()
    t_fb = timeit(fb, iters=5, warm=1)
    print('[%d,%d] %s: abs-max %.3f ms | AbsPercentile(99.999) forward %.3f ms (%.1f x abs-max) | forward + backward %.3f ms' % (
        outer, ch, str(dt).replace('torch.', ''), t_abs, t_sel, t_sel / t_abs, t_fb))
PY
