"""N > 1 path on CPU: world_size-2 gloo run of the sharding protocol (brevitas_amd.distributed).

The collective logic is device-agnostic torch code; the per-shard numbers it exchanges are produced
here by the oracle (on the GPU box they come from the HIP kernels, covered by the -m gpu tests).
Property checked: batch-sharded == single process on the concatenated batch --
  statistic and scale identical, y of each shard identical to the matching slice (bit-exact),
  scale-gradient sums equal (double accumulation, fixed rank order: identical on both ranks),
  and exactly one shard -- the one holding the first arg-max in batch order -- keeps each
  channel's deposit.
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from mp_util import init_gloo, run_ranks


def _first_positions(x_shard, stat, c, inner):
    """what the backward kernel records: first (outer*inner + i) position with |x| == stat[c], or -1"""
    n = x_shard.shape[0]
    xs = np.abs(x_shard.reshape(n, c, inner))
    out = np.full(c, -1, dtype=np.int64)
    for ch in range(c):
        hit = np.nonzero(xs[:, ch, :].reshape(-1) == stat[ch])[0]
        if hit.size:
            out[ch] = hit[0]
    return out


def _worker(rank, world, port, per_channel, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import oracle as O
    from brevitas_amd.distributed import sync_backward, sync_stat_max
    init_gloo(rank, world, port)
    try:
        g = torch.Generator().manual_seed(123456)
        n, c, h, w = 4, 6, 5, 4
        x = torch.randn(n, c, h, w, generator=g)
        x[3, 2, 1, 1] = 9.0  # channel 2's maximum lives in the second shard
        x[0, 4, 0, 0] = -7.5
        x[2, 4, 2, 2] = 7.5  # a +-max tie across shards: the first (rank 0) must own it
        gr = torch.randn(n, c, h, w, generator=g)
        inner = h * w
        per = n // world
        xs, gs = x[rank * per:(rank + 1) * per], gr[rank * per:(rank + 1) * per]
        xn, gn = xs.reshape(-1).numpy().copy(), gs.reshape(-1).numpy().copy()
        ch = c if per_channel else 1
        lay_s = (per, c, inner) if per_channel else (1, 1, xn.size)
        lay_f = (n, c, inner) if per_channel else (1, 1, x.numel())

        # forward: local statistic -> all-reduce(MAX) -> same scale everywhere
        stat_local = O.stats(O.STAT_ABSMAX, xn, O.F32, *lay_s)
        stat = sync_stat_max(torch.from_numpy(stat_local.copy()), dist.group.WORLD).numpy()
        stat_full = O.stats(O.STAT_ABSMAX, x.reshape(-1).numpy().copy(), O.F32, *lay_f)
        assert np.array_equal(stat, stat_full)
        scale = (np.maximum(stat, np.float32(1e-10)) / np.float32(128.0)).astype(np.float32)
        zp = np.zeros(1, dtype=np.float32)
        d_s = O.make_desc(*lay_s, O.F32, O.F32, O.F32, scale_per_channel=per_channel, qmin=-128.0, qmax=127.0)
        d_f = O.make_desc(*lay_f, O.F32, O.F32, O.F32, scale_per_channel=per_channel, qmin=-128.0, qmax=127.0)
        y_s, _ = O.fakequant_fwd(d_s, xn, scale, zp)
        y_f, _ = O.fakequant_fwd(d_f, x.reshape(-1).numpy().copy(), scale, zp)
        assert np.array_equal(y_s, y_f.reshape(n, -1)[rank * per:(rank + 1) * per].reshape(-1))

        # backward: local sums + local tie bookkeeping -> one all-gather
        _, ds_local, _ = O.fakequant_bwd(d_s, gn, xn, scale, zp)
        _, ds_full, _ = O.fakequant_bwd(d_f, gr.reshape(-1).numpy().copy(), x.reshape(-1).numpy().copy(), scale, zp)
        if per_channel:
            info = torch.from_numpy(_first_positions(xs.numpy(), stat, c, inner))
        else:
            cnt = int(np.count_nonzero(np.abs(xn) == stat[0]))
            info = torch.tensor([cnt, 0], dtype=torch.int64)
        ds_total, info2, total = sync_backward(torch.from_numpy(ds_local.copy()), info, ch, dist.group.WORLD)
        np.testing.assert_allclose(ds_total.numpy(), ds_full, rtol=1e-6, atol=1e-6)
        gathered = [torch.zeros_like(ds_total) for _ in range(world)]
        dist.all_gather(gathered, ds_total)
        assert all(torch.equal(gathered[0].view(torch.int32), t.view(torch.int32)) for t in gathered)
        if per_channel:
            full_first = _first_positions(x.numpy(), stat, c, inner)
            owner_rank = full_first // (per * inner)
            keep = info2[:c].numpy() >= 0
            assert np.array_equal(keep, owner_rank == rank), (rank, keep, owner_rank)
            # the kept local position is the global first position, shifted by the shard offset
            assert np.array_equal(info2[:c].numpy()[keep] + rank * per * inner, full_first[keep])
            assert total is None
        else:
            full_cnt = int(np.count_nonzero(np.abs(x.numpy()) == stat[0]))
            assert int(total) == full_cnt
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('per_channel', [True, False], ids=['per_channel', 'per_tensor'])
def test_sharded_equals_full_batch(oracle, per_channel):
    run_ranks(_worker, 2, per_channel, timeout=180)
