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


def _check_dx(got, want, max_deposits, dtype, what):
    """dx of a shard against the slice of the full-batch dx: bit-exact except at the elements that receive a statistic's
    gradient -- the shards' scale-gradient sums are rounded before they are added, the full batch rounds once -- and at
    most `max_deposits` of those over BOTH shards (a deposit lands on exactly one of them)"""
    diff = (_bits(got) != _bits(want)).nonzero()
    ndiff = torch.tensor([diff.shape[0]])
    dist.all_reduce(ndiff)
    assert int(ndiff) <= max_deposits, (what, 'differing elements over both shards', int(ndiff))
    tol = 0.05 if dtype == torch.bfloat16 else 1e-4
    g, w = got.float(), want.float()
    assert torch.allclose(g, w, rtol=tol, atol=tol * float(w.abs().max())), what


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
                _check_dx(xs.grad, xf.grad[lo:hi], c if per_channel else 1, dtype, ('dx', dtype, per_channel))
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
                _check_dx(xs.grad, xf.grad[rank * 4:(rank + 1) * 4], 1, dtype, ('percentile dx', dtype, step))
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
