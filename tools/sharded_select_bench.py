"""Developer tool: the sharded percentile of a whole-tensor statistic, per shard -- 11-bit digit passes (bvq_kth_*)
against the 15-bit first digit (bvq_kthw_*), on the headline tensor as ONE shard; the all-reduces between hist and
pick are left out (they move channels*2048*4 B = 8 KB per pass, or 128 KB + 256/512 KB)."""
import statistics
import sys

import torch

sys.path.insert(0, '.')
from brevitas_amd import _native as nat  # noqa: E402


def run(steps):
    steps.begin()
    for p in range(steps.passes):
        steps.hist(p)
        steps.pick(p)
    return steps.finish()


def timed(fn, rounds=9):
    fn()
    fn()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    n = 256 * 512 * 56 * 56
    for name, dt in (('bf16', torch.bfloat16), ('f16', torch.float16), ('f32', torch.float32)):
        x = torch.randn(n, device='cuda:0', dtype=dt)
        for abs_key, what in ((True, 'AbsPercentile(99.999)'), (False, 'NegativePercentileOrZero(0.001)')):
            rule, q = (nat.KTH_HIGH, 99.999) if abs_key else (nat.KTH_LOW, 0.001)
            a = nat.KthSelectSteps(x, 1, 1, n, abs_key, rule, q)
            b = nat.KthWideSteps(x, abs_key, rule, q)
            assert torch.equal(run(a), run(b))
            ta, tb = timed(lambda: run(a)), timed(lambda: run(b))
            print('%-4s %-32s 11-bit passes (%d reads): %.3f ms | 15-bit first digit (%d reads): %.3f ms' % (
                name, what, a.passes, ta, b.passes, tb), flush=True)
        del x


if __name__ == '__main__':
    main()
