"""Weight-sized quantizer steps (configs 2, 4, 5): eager, HIP-graph replay of the whole step, and
torch.cuda.make_graphed_callables -- the route that keeps autograd in the loop (the quantizer's forward and backward
each become ONE graph launch inside an ordinary training step).  us per forward + backward."""
import sys
import time

import torch

sys.path.insert(0, '.')
import brevitas_amd.quant as Q  # noqa: E402


def timeit(fn, iters=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = 'cuda:0'
    for shape, dt, bits in (((512, 512, 3, 3), torch.float32, 8), ((256, 1024, 1, 1), torch.bfloat16, 8),
                            ((1024, 256, 1, 1), torch.bfloat16, 8), ((8192, 8192), torch.bfloat16, 4)):
        torch.manual_seed(0)
        w = torch.nn.Parameter((torch.randn(shape, device=dev) * 0.02).to(dt))
        g = torch.randn(shape, device=dev, dtype=dt)
        q = Q.Int8WeightPerChannelFloat(w, bit_width=bits).to(dev)

        def eager():
            w.grad = None
            q(w)[0].backward(g)

        t_eager = timeit(eager)
        ref = w.grad.clone()
        # the quantizer as a graphed callable: forward and backward are one graph launch each
        qg = torch.cuda.make_graphed_callables(q, (w,))

        def graphed():
            w.grad = None
            qg(w)[0].backward(g)

        t_graphed = timeit(graphed)
        same = torch.equal(w.grad, ref)
        print('%-20s %-8s int%d  eager %7.1f us   make_graphed_callables %7.1f us   (gradients identical: %s)' % (
            list(shape), str(dt)[6:], bits, t_eager, t_graphed, same), flush=True)


if __name__ == '__main__':
    main()
