"""The batch-sharded path with TWO ranks on device tensors: both processes use the box's one GPU and exchange
through gloo (RCCL refuses two ranks on one device), so every device-side branch of brevitas_amd.distributed --
the statistic's all-reduce on a device tensor, bvq_shard_pack / all-gather / bvq_shard_unpack, the deposit on
the owning shard, the sharded radix select -- runs with world size 2 on real kernels.  Reference: the same
quantizer, unsharded, on the concatenated batch (computed by rank 0 on the same device)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bits(t):
    return t.detach().contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32)


def _check_dx(got, want, x_shard, g_shard, g_full, stat, per_channel, max_ulp, what):
    """dx of a shard against the slice of the full-batch dx: bit-exact everywhere except at elements that RECEIVE the
    statistic's gradient, i.e. elements attaining the statistic (|x| == stat of their channel).  There both runs add
    float32 partial sums of the scale gradient in double -- since round 3 the shards' sums are no longer rounded to
    float32 before they are added -- but the partials themselves come from different unit decompositions (8 rows vs 4
    rows per channel), and a float32 partial carries an error of one unit in the last place of the MAGNITUDES it
    summed (sum |g| over the channel, / int_threshold once it is the statistic's gradient).  So the two dx values
    differ by at most `max_ulp` units in the last place of max(deposited term, that magnitude); the term is read off
    want - g (dx = own + term with own ~ g).  (Round 2 tolerated 5 % of the largest |dx| and never looked at the
    positions.)"""
    diff = _bits(got) != _bits(want)
    ax = x_shard.detach().abs()
    st = stat.detach().to(ax.dtype)
    attains = ax == (st.reshape(1, -1, 1, 1) if per_channel else st.reshape(()))
    assert not bool((diff & ~attains).any()), (what, 'a differing element does not attain the statistic')
    if bool(diff.any()):
        summed = g_full.double().abs().sum(dim=(0, 2, 3) if per_channel else None) / 128.0
        summed = (summed.reshape(1, -1, 1, 1) if per_channel else summed.reshape(1, 1, 1, 1)).expand_as(got)[diff]
        gd, wd, og = got[diff].double(), want[diff].double(), g_shard[diff].double()
        mag = torch.maximum(torch.maximum(gd.abs(), wd.abs()), torch.maximum((wd - og).abs(), summed))
        mant = 7 if got.dtype == torch.bfloat16 else 23
        ulp = torch.exp2(torch.floor(torch.log2(mag)) - mant)
        worst = float(((gd - wd).abs() / ulp).max())
        assert worst <= max_ulp, (what, 'ulps between the two runs at a deposit', worst)
    ndiff = torch.tensor([int(diff.sum())])
    dist.all_reduce(ndiff)
    return int(ndiff)


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        import brevitas_amd.quant as Q
        from bench import build_quantizer
        from brevitas_amd.distributed import shard_over_batch
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        group = dist.group.WORLD
        for dtype in (torch.bfloat16, torch.float32):
            for per_channel in (True, False):
                gen = torch.Generator().manual_seed(123456)
                n, c, h, w = 8, 16, 14, 14
                x = torch.randn(n, c, h, w, generator=gen).to(dtype)
                x[1, 3, 2, 2] = 7.0
                x[6, 3, 1, 1] = -7.0   # channel 3: a +-max tie ACROSS the two shards -- the first one owns the deposit
                x[5, 9, 0, 0] = 9.5    # channel 9's maximum lives in the second shard
                g = torch.randn(n, c, h, w, generator=gen).to(dtype)
                per = n // world
                xs = x[rank * per:(rank + 1) * per].to(dev).requires_grad_(True)
                qs = build_quantizer(c, per_channel, dev, group)
                y, scale, _, _ = qs(xs)
                y.backward(g[rank * per:(rank + 1) * per].to(dev))
                # the unsharded quantizer on the whole batch, same device
                xf = x.to(dev).requires_grad_(True)
                qf = build_quantizer(c, per_channel, dev)
                yf, scalef, _, _ = qf(xf)
                yf.backward(g.to(dev))
                torch.cuda.synchronize()
                lo, hi = rank * per, (rank + 1) * per
                assert torch.equal(_bits(scale), _bits(scalef)), ('scale', dtype, per_channel)
                assert torch.equal(_bits(y), _bits(yf[lo:hi])), ('y', dtype, per_channel)
                stat = scalef.float() * 128.0   # the statistic itself (powers of two scale exactly)
                nd = _check_dx(xs.grad, xf.grad[lo:hi], xs, g[lo:hi].to(dev), g.to(dev), stat, per_channel, 1 if dtype == torch.bfloat16 else 4,
                               ('dx', dtype, per_channel))
                assert nd <= (c if per_channel else 1), ('deposits that differ', nd)
                assert torch.equal(_bits(qs.scaling_impl.runtime_stats.running_stats),
                                   _bits(qf.scaling_impl.runtime_stats.running_stats)), ('running', dtype, per_channel)
            # the default activation quantizer: percentile statistic (sharded radix select, 15-bit first digit) over its
            # collection steps, the switch to the learned scale, one learned step
            qa = Q.Int8ActPerTensorFloat(collect_stats_steps=2).to(dev)
            qb = shard_over_batch(Q.Int8ActPerTensorFloat(collect_stats_steps=2).to(dev), group)
            qa.train(), qb.train()
            for step in range(4):
                gen = torch.Generator().manual_seed(1000 + step)
                x = (torch.randn(8, 16, 14, 14, generator=gen) * (1.0 + step)).to(dtype)
                g = torch.randn(8, 16, 14, 14, generator=gen).to(dtype)
                xf = x.to(dev).requires_grad_(True)
                xs = x[rank * 4:(rank + 1) * 4].to(dev).requires_grad_(True)
                ya, sa = qa(xf)[:2]
                yb, sb = qb(xs)[:2]
                ya.backward(g.to(dev))
                yb.backward(g[rank * 4:(rank + 1) * 4].to(dev))
                torch.cuda.synchronize()
                assert torch.equal(_bits(sa), _bits(sb)), ('percentile scale', dtype, step)
                assert torch.equal(_bits(yb), _bits(ya[rank * 4:(rank + 1) * 4])), ('percentile y', dtype, step)
                # collection steps: the k-th value's gradient lands on the element(s) holding it; afterwards the scale
                # is a learned parameter and dx must be bit-identical
                nd = _check_dx(xs.grad, xf.grad[rank * 4:(rank + 1) * 4], xs, g[rank * 4:(rank + 1) * 4].to(dev), g.to(dev), sa.float() * 128.0, False,
                               1 if dtype == torch.bfloat16 else 4, ('percentile dx', dtype, step))
                assert nd <= (1 if step < 2 else 0), ('percentile deposits that differ', step, nd)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_device_equal_the_full_batch():
    world = 2
    ctx = mp.get_context('forkserver')  # started by conftest.py before anything touched the GPU
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:  # a worker that is still alive is stuck: do not leave it on the GPU
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)
    results = [q.get(timeout=5) for _ in range(world)]
    for rank, msg in results:
        assert msg == 'ok', 'rank %d failed:\n%s' % (rank, msg)
    assert all(p.exitcode == 0 for p in procs)
