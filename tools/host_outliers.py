"""Developer tool: per-call host time of a weight-sized quantizer step -- what is the slow mode seen at the start of a run?

    python tools/host_outliers.py        (GPU box)

2000 calls of `w.grad = None; q(w)[0].backward(g)` per series: median host time of the forward and of the backward call,
the GPU time the same calls took (two HIP events around the series), after different kinds of pause."""
import statistics
import sys
import time

import torch

sys.path.insert(0, '.')
import brevitas_amd.quant as Q  # noqa: E402


def main():
    dev = 'cuda:0'
    w = torch.nn.Parameter(torch.randn(512, 512, 3, 3, device=dev) * 0.02)
    g = torch.randn(512, 512, 3, 3, device=dev)
    q = Q.Int8WeightPerChannelFloat(w).to(dev)
    big = torch.randn(64 * 1024 * 1024, device=dev)

    def series(name, n=2000, reset=True):
        for _ in range(20):
            w.grad = None
            q(w)[0].backward(g)
        torch.cuda.synchronize()
        tf, tb = [], []
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        t_start = time.perf_counter()
        for _ in range(n):
            t0 = time.perf_counter()
            if reset:
                w.grad = None
            y = q(w)[0]
            t1 = time.perf_counter()
            y.backward(g)
            t2 = time.perf_counter()
            tf.append((t1 - t0) * 1e6)
            tb.append((t2 - t1) * 1e6)
        host = time.perf_counter() - t_start
        e1.record()
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        drain = time.perf_counter() - t0
        k = n // 4
        print('%-44s host %6.1f us/call (fwd %5.1f, bwd %5.1f median; first quarter %6.1f, last quarter %6.1f)  '
              'GPU span %6.1f us/call  drain %5.0f us' % (
                  name, host / n * 1e6, statistics.median(tf), statistics.median(tb),
                  statistics.mean([a + b for a, b in zip(tf[:k], tb[:k])]),
                  statistics.mean([a + b for a, b in zip(tf[-k:], tb[-k:])]),
                  e0.elapsed_time(e1) * 1e3 / n, drain * 1e6))

    series('A: first thing in the process')
    series('B: straight after A')
    time.sleep(1.0)
    series('C: after time.sleep(1.0)')
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        pass
    series('D: after 1 s of host spinning, GPU idle')
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        big.add_(1.0)
    torch.cuda.synchronize()
    series('E: after 1 s of large GPU kernels')
    series('F: straight after E')
    series('G: accumulate (no w.grad reset)', reset=False)
    series('H: 10000 calls', n=10000)


if __name__ == '__main__':
    main()
