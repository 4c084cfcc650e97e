"""Batch-sharded quantizers on CPU tensors (world size 2, gloo): the module surface with `shard_over_batch` through the
package's pure-torch CPU route (brevitas_amd/_aten.py: all-reduce(MAX) of the statistic, the gradient routed to the shard
that owns the arg-max).  Property: sharded == one process on the concatenated batch -- scale, running statistics and y
bit-identical; dx bit-identical away from the elements that receive the statistic's gradient, which is the SUM over
the shards of per-shard sums (each rounded to the compute dtype by autograd, as the reference's ops do) and so agrees
with the one-process sum to a few ulps."""
import os

import pytest
import torch
import torch.distributed as dist

from mp_util import init_gloo, run_ranks


def _worker(rank, world, port, per_channel, stat_kind, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    init_gloo(rank, world, port)
    try:
        from bench import build_quantizer
        from brevitas_amd.core.stats import AbsMinMax
        from brevitas_amd.distributed import shard_over_batch
        g = torch.Generator().manual_seed(123456)
        n, c, h, w = 4, 6, 5, 4
        x = torch.randn(n, c, h, w, generator=g)
        x[3, 2, 1, 1] = 9.0    # channel 2's maximum lives in the second shard
        x[0, 4, 0, 0] = -7.5
        x[2, 4, 2, 2] = 7.5    # a +-max tie across the shards: the first in batch order (rank 0) owns it
        gr = torch.randn(n, c, h, w, generator=g)
        per = n // world
        sl = slice(rank * per, (rank + 1) * per)

        def make(group):
            qz = build_quantizer(c, per_channel, torch.device('cpu'))
            if stat_kind == 'minmax':
                qz.scaling_impl.runtime_stats.stats.stats_impl = AbsMinMax(1 if per_channel else None)
            if group is not None:
                shard_over_batch(qz, group)
            return qz
        # (native collectives are RCCL's: on a gloo group every rank is told so and nothing changes)
        from brevitas_amd.distributed import enable_native_collectives
        assert enable_native_collectives() is False
        full, shard = make(None), make(dist.group.WORLD)
        for step in range(2):   # the second step also exercises the running-statistics update
            xf = x.clone().requires_grad_(True)
            xs = x[sl].clone().requires_grad_(True)
            yf, sf = full(xf)[:2]
            ys, ss = shard(xs)[:2]
            yf.backward(gr)
            ys.backward(gr[sl])
            assert torch.equal(sf, ss), (step, 'scale')
            assert torch.equal(ys, yf[sl]), (step, 'y')
            rf = full.scaling_impl.runtime_stats.running_stats
            rs = shard.scaling_impl.runtime_stats.running_stats
            assert torch.equal(rf, rs), (step, 'running statistics')
            d_full, d_shard = xf.grad[sl], xs.grad
            diff = (d_full != d_shard).nonzero()
            stat = (sf * 128.0)
            # every differing position holds an element attaining the statistic
            for idx in diff:
                i, ch = int(idx[0]), int(idx[1])
                if stat_kind == 'absmax':
                    s = stat.reshape(-1)[ch if per_channel else 0]
                    assert abs(float(xs.detach()[tuple(idx)])) == pytest.approx(float(s), rel=1e-6), (step, idx)
            limit = 2 * (c if per_channel else int((x.abs() == x.abs().max()).sum()))
            assert diff.shape[0] <= limit, (step, diff.shape[0])
            assert torch.allclose(d_full, d_shard, rtol=1e-5, atol=1e-5 * float(xf.grad.abs().max())), (step, 'dx')
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('stat_kind', ['absmax', 'minmax'])
@pytest.mark.parametrize('per_channel', [True, False], ids=['per_channel', 'per_tensor'])
def test_sharded_cpu_route_equals_full_batch(per_channel, stat_kind):
    run_ranks(_worker, 2, per_channel, stat_kind, timeout=180)
