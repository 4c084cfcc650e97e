"""Developer tool: where does the HOST time of one small quantizer step go? (cProfile on the GPU box)"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, '.')
import brevitas_amd.quant as Q  # noqa: E402


def main():
    dev = 'cuda:0'
    w = torch.nn.Parameter(torch.randn(512, 512, 3, 3, device=dev) * 0.02)
    g = torch.randn(512, 512, 3, 3, device=dev)
    q = Q.Int8WeightPerChannelFloat(w).to(dev)

    def step():
        w.grad = None
        y = q(w)[0]
        y.backward(g)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)


if __name__ == '__main__':
    main()
