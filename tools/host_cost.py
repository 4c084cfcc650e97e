"""Developer tool: host microseconds of one weight-sized quantizer step, split by layer (run on the GPU box).

    python tools/host_cost.py [conv|linear]

The kernels of a [512,512,3,3] weight take ~35 us per step; everything above that is Python, ctypes and autograd.
Each line is the wall time per call of a piece of the step, measured over 2000 calls with the GPU kept busy but
never waited for (the stream runs ahead of nothing: pieces that launch are launch-rate bound at worst)."""
import sys
import time

import torch

sys.path.insert(0, '.')
import brevitas_amd.quant as Q  # noqa: E402
from brevitas_amd import _native as nat  # noqa: E402
from brevitas_amd.core.quant import _fused  # noqa: E402


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


class _Twice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return nat.unary(nat.OP_ABS, x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return nat.unary(nat.OP_ABS, g)


def main():
    dev = 'cuda:0'
    which = sys.argv[1] if len(sys.argv) > 1 else 'conv'
    shape, dt, bits = ((512, 512, 3, 3), torch.float32, 8) if which == 'conv' else ((8192, 8192), torch.bfloat16, 4)
    w = torch.nn.Parameter((torch.randn(shape, device=dev) * 0.02).to(dt))
    g = torch.randn(shape, device=dev, dtype=dt)
    q = Q.Int8WeightPerChannelFloat(w, bit_width=bits).to(dev)
    tq = q  # the factory returns the tensor_quant module itself
    rows = []

    def step():
        w.grad = None
        q(w)[0].backward(g)
    rows.append(('full step: w.grad = None; q(w)[0].backward(g)', per_call(step)))

    def fwd_grad():
        return q(w)[0]
    rows.append(('forward, autograd recording', per_call(fwd_grad)))

    def fwd_nograd():
        with torch.no_grad():
            return q(w)[0]
    rows.append(('forward under no_grad', per_call(fwd_nograd)))
    y = q(w)[0]

    def bwd_only():
        w.grad = None
        y.backward(g, retain_graph=True)
    rows.append(('backward alone (retain_graph)', per_call(bwd_only)))

    # the floor: the same shape of step on stock torch, and on a do-nothing custom Function with one launch each way
    def floor_torch():
        w.grad = None
        torch.mul(w, 2.0).backward(g)
    rows.append(('floor: torch.mul(w, 2).backward(g)', per_call(floor_torch)))

    def floor_fn():
        w.grad = None
        _Twice.apply(w).backward(g)
    rows.append(('floor: custom Function, one C-ABI launch each way', per_call(floor_fn)))

    class _Shape(torch.autograd.Function):  # the quantizer Function's signature, one launch each way
        @staticmethod
        def forward(ctx, x, a, b, c, d, e, f, h, i, j):
            ctx.set_materialize_grads(False)
            y = nat.unary(nat.OP_ABS, x)
            s1 = torch.empty(x.shape[0], device=x.device)
            s2 = torch.empty(x.shape[0], device=x.device)
            ctx.save_for_backward(x, s1, s2, a)
            s2v = s2.view(-1, 1, 1, 1) if x.dim() == 4 else s2.view(-1, 1)
            ctx.mark_non_differentiable(s2v)
            return y, (s1.view(-1, 1, 1, 1) if x.dim() == 4 else s1.view(-1, 1)), s2v

        @staticmethod
        def backward(ctx, g, gs, _g2):
            x, s1, s2, a = ctx.saved_tensors
            return nat.unary(nat.OP_ABS, g), None, None, None, None, None, None, None, None, None

    bw0 = tq.msb_clamp_bit_width_impl()
    thr0 = tq.int_scaling_impl(bw0)

    def floor_shape():
        w.grad = None
        _Shape.apply(w, thr0, None, -128.0, 127.0, 0, False, None, 0, None)[0].backward(g)
    rows.append(('floor: custom Function with the quantizer\'s signature (10 inputs, 3 outputs, 4 saved)', per_call(floor_shape)))

    def step_accumulate():
        q(w)[0].backward(g)
    rows.append(('full step without resetting w.grad (accumulates)', per_call(step_accumulate)))

    def step_grad():
        return torch.autograd.grad(q(w)[0], w, g)
    rows.append(('full step through torch.autograd.grad (no AccumulateGrad)', per_call(step_grad)))

    # pieces of the module forward
    bw = tq.msb_clamp_bit_width_impl()
    rows.append(('  msb_clamp_bit_width_impl()', per_call(lambda: tq.msb_clamp_bit_width_impl())))
    rows.append(('  _stats_plan(x, bit_width)', per_call(lambda: tq._stats_plan(w, bw))))
    rows.append(('  int_scaling_impl(bit_width)', per_call(lambda: tq.int_scaling_impl(bw))))
    sp, tmpl = tq._stats_plan(w, bw)
    thr = tq.int_scaling_impl(bw)

    def apply_only():
        return _fused.StatsFakeQuantFn.apply(w, thr, sp, tmpl['qmin'], tmpl['qmax'], tmpl['round_mode'],
                                             tmpl['clamp_ste'], None, nat.PRE_NONE, None)
    rows.append(('  StatsFakeQuantFn.apply (forward, recording)', per_call(apply_only)))
    def fn_step():
        w.grad = None
        apply_only()[0].backward(g)
    rows.append(('  StatsFakeQuantFn.apply(...)[0].backward(g) (no module layers)', per_call(fn_step)))
    yy, scale, stat = apply_only()
    rows.append(('  zero_point_impl(x, scale, bit_width)', per_call(lambda: tq.zero_point_impl(w, scale, bw))))

    # the C-ABI wrappers alone
    code = nat.dtype_code(dt)
    outer, ch, inner = 1, shape[0], w.numel() // shape[0]
    desc = nat.QuantDesc(outer, ch, inner, code, code, code, nat.F32, 1, 0, float(-2 ** (bits - 1)),
                         float(2 ** (bits - 1) - 1), 0, 0, 0, nat.OUT_DEQUANT, nat.PRE_NONE)
    zp = torch.zeros(1, device=dev)
    flat = w.detach().reshape(-1)
    thr_div = float(2 ** (bits - 1))
    r = nat.stats_fakequant_fwd(desc, flat, None, thr_div, dt)
    if r is not None:
        rows.append(('  nat.stats_fakequant_fwd (3 allocations + 2 ctypes calls)',
                     per_call(lambda: nat.stats_fakequant_fwd(desc, flat, None, thr_div, dt))))
        st, sc, _ = r
    else:
        st, sc = nat.absmax_scale(flat, outer, ch, inner, None, thr_div, dt)
        rows.append(('  nat.absmax_scale + nat.fakequant_fwd', per_call(
            lambda: (nat.absmax_scale(flat, outer, ch, inner, None, thr_div, dt), nat.fakequant_fwd(desc, flat, sc, zp)))))
    gf = g.reshape(-1)
    rows.append(('  nat.fakequant_bwd_stats (3 allocations + 2 ctypes calls)',
                 per_call(lambda: nat.fakequant_bwd_stats(desc, gf, flat, sc, zp, st, dt, thr_div, dt))))
    rows.append(('  torch.empty_like(w)', per_call(lambda: torch.empty_like(w))))
    rows.append(('  nat.QuantDesc(...) construction', per_call(lambda: nat.QuantDesc(
        outer, ch, inner, code, code, code, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, nat.OUT_DEQUANT, nat.PRE_NONE))))
    rows.append(('full step again, last (the process is warm now)', per_call(step)))
    rows.append(('full step again, 10000 calls', per_call(step, 10000)))
    print('%s %s int%d' % (list(shape), str(dt)[6:], bits))
    for name, us in rows:
        print('%8.1f us  %s' % (us, name))


if __name__ == '__main__':
    main()
