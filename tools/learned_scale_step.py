"""Steady state of the default activation quantizer (Int8ActPerTensorFloat after its collection phase: learned
scale): time per step and the launches it consists of (run under rocprofv3 --kernel-trace for the breakdown)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import brevitas_amd.quant as Q  # noqa: E402


def main():
    dev = 'cuda:0'
    dt = torch.bfloat16 if 'f32' not in sys.argv else torch.float32
    x = torch.randn(256, 512, 56, 56, device=dev, dtype=dt).requires_grad_(True)
    g = torch.randn(256, 512, 56, 56, device=dev, dtype=dt)
    q = Q.Int8ActPerTensorFloat(collect_stats_steps=3).to(dev)
    q.train()

    def step():
        x.grad = None
        for p in q.parameters():
            p.grad = None
        y = q(x)[0]
        y.backward(g)

    for _ in range(60):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 100
    for _ in range(n):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print('Int8ActPerTensorFloat steady state (learned scale) %s [256,512,56,56]: %.3f ms / step = %.0f Gelem/s' % (
        str(dt)[6:], ms, x.numel() / ms / 1e6))


if __name__ == '__main__':
    main()
