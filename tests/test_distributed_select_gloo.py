"""Distributed percentile (radix select) protocol on CPU: world_size-2 gloo run of
brevitas_amd.distributed.sharded_kth_value and of the first-element ownership exchange.

The protocol code is device-agnostic; the per-shard digit histograms it sums are produced here by a
numpy restatement of the select steps (on the GPU box they come from bvq_kth_hist / bvq_kth_pick,
covered by the -m gpu tests).  Property checked: the value selected over two shards equals
torch.kthvalue on the concatenated tensor, for the rank rules of AbsPercentile (floor(.01 q n + .5)) and
NegativePercentileOrZero (ceil(.01 q n)) evaluated from the GLOBAL count, for unequal shard sizes, ties
and an empty shard; and exactly one shard -- the first in batch order that holds the value -- keeps the
gradient deposit."""
import math
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from mp_util import init_gloo, run_ranks

KBITS = 11


def _keys(x, abs_key):
    """order-preserving unsigned keys of float32 values (include/bvq.h, bvq_kth_value)"""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    if abs_key:
        return b & 0x7fffffff
    neg = (b >> 31) == 1
    return np.where(neg, b ^ 0xffffffff, b ^ 0x80000000)


def _unkey(k, abs_key):
    k = np.uint64(k)
    if abs_key:
        b = k
    else:
        b = k ^ np.uint64(0x80000000) if (k >> np.uint64(31)) else k ^ np.uint64(0xffffffff)
    return np.array([b], dtype=np.uint64).astype(np.uint32).view(np.float32)[0]


class NumpySelectSteps:
    """stand-in for brevitas_amd._native.KthSelectSteps over one float32 shard (one channel)"""
    passes = 3

    def __init__(self, x, abs_key, rule, q):
        self.keys, self.abs_key, self.rule, self.q = _keys(x.reshape(-1), abs_key), abs_key, rule, q

    def begin(self):
        self.prefix, self.done, self.k = 0, 0, None
        self.h = None

    def _bits(self, p):
        return KBITS if p < 2 else 10

    def hist(self, p):
        bits = self._bits(p)
        shift = 32 - self.done - bits
        sel = self.keys if p == 0 else self.keys[(self.keys >> np.uint64(shift + bits)) == np.uint64(self.prefix)]
        digits = ((sel >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        counts = np.bincount(digits, minlength=1 << KBITS).astype(np.uint32)
        self.h = torch.from_numpy(counts.view(np.int32).copy())
        return self.h

    def pick(self, p):
        counts = self.h.numpy().view(np.uint32).astype(np.int64)
        if p == 0:
            n = int(counts.sum())
            v = .01 * self.q * n
            self.k = int(math.floor(v + 0.5)) if self.rule == 1 else int(math.ceil(v))
            self.k = min(max(self.k, 1), n)
        cum = np.cumsum(counts)
        d = int(np.searchsorted(cum, self.k, side='left'))
        self.k -= int(cum[d - 1]) if d else 0
        bits = self._bits(p)
        self.prefix = (self.prefix << bits) | d
        self.done += bits

    def finish(self):
        return torch.tensor([_unkey(self.prefix, self.abs_key)])


class NumpyWideSteps(NumpySelectSteps):
    """stand-in for brevitas_amd._native.KthWideSteps over one float32 shard: a 15-bit first digit, then the
    remaining 16 (|x|) or 17 bits in ONE pass -- the histogram the shards sum has a different size per pass"""

    def __init__(self, x, abs_key, rule, q):
        super().__init__(x, abs_key, rule, q)
        self.key_bits = 31 if abs_key else 32
        self.passes = 2

    def _bits(self, p):
        return 15 if p == 0 else self.key_bits - 15

    def hist(self, p):
        bits = self._bits(p)
        shift = self.key_bits - self.done - bits
        sel = self.keys if p == 0 else self.keys[(self.keys >> np.uint64(shift + bits)) == np.uint64(self.prefix)]
        digits = ((sel >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        counts = np.bincount(digits, minlength=1 << bits).astype(np.uint32)
        self.h = torch.from_numpy(counts.view(np.int32).copy())
        return self.h


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from brevitas_amd.distributed import sharded_kth_value, sync_backward
    init_gloo(rank, world, port)
    try:
        g = torch.Generator().manual_seed(123456)
        full = torch.randn(1000, generator=g)
        full[100:140] = 0.75   # a run of ties straddling nothing, inside shard 0
        full[700:705] = -0.75  # |x| ties in shard 1
        full[[3, 900]] = 2.5   # the same value in both shards
        for split in (600, 1, 0, 1000):  # unequal shards, a one-element shard, an empty shard on either side
            mine = (full[:split] if rank == 0 else full[split:]).numpy()
            for abs_key in (True, False):
                for rule, qs in ((1, (99.999, 50.0, 90.0, 0.2, 100.0)), (2, (0.001, 10.0, 33.3, 100.0))):
                    for qq in qs:
                        steps = NumpySelectSteps(mine, abs_key, rule, qq)
                        got = float(sharded_kth_value(steps, dist.group.WORLD))
                        wide = float(sharded_kth_value(NumpyWideSteps(mine, abs_key, rule, qq), dist.group.WORLD))
                        assert wide == got, (split, abs_key, rule, qq, wide, got)
                        n = full.numel()
                        k = int(math.floor(.01 * qq * n + 0.5)) if rule == 1 else int(math.ceil(.01 * qq * n))
                        src = full.abs() if abs_key else full
                        want = float(src.kthvalue(k).values)
                        assert got == want, (split, abs_key, rule, qq, got, want)
        # ownership of the single deposit (kthvalue's gradient): value 2.5 lives at 3 (rank 0) and 900 (rank 1)
        split = 600
        lo, hi = (0, split) if rank == 0 else (split, 1000)
        hit = np.nonzero(full[lo:hi].numpy() == 2.5)[0]
        info = torch.tensor([hit[0] if hit.size else -1, 0], dtype=torch.int64)
        gval = torch.tensor([1.0 + rank])  # every shard back-propagates its own gradient of the statistic
        gsum, info2, total = sync_backward(gval, info, 1, dist.group.WORLD, first_only=True)
        assert total is None and float(gsum) == 3.0
        assert (int(info2[0]) >= 0) == (rank == 0)  # the first shard in batch order keeps it
        # a value only the second shard holds
        hit = np.nonzero(full[lo:hi].numpy() == -0.75)[0]
        info = torch.tensor([hit[0] if hit.size else -1, 0], dtype=torch.int64)
        gsum, info2, total = sync_backward(gval, info, 1, dist.group.WORLD, first_only=True)
        assert (int(info2[0]) >= 0) == (rank == 1)
        q.put((rank, 'ok'))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_select_equals_kthvalue_of_concatenation():
    run_ranks(_worker, 2, timeout=240)
