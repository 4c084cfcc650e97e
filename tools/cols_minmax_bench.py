"""min/max statistic on channel-last layouts: column-mapped route against the row-mapped one (misaligned copy)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brevitas_amd import _native as nat

def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for shape, dt in (((65536, 4096, 1), torch.bfloat16), ((256 * 56 * 56, 512, 1), torch.bfloat16),
                  ((8192, 1024, 49), torch.bfloat16), ((65536, 4096, 1), torch.float32)):
    outer, ch, inner = shape
    x = torch.randn(outer * ch * inner + 16, device='cuda:0', dtype=dt)
    al, mis = x[:outer * ch * inner], x[1:1 + outer * ch * inner]
    assert al.data_ptr() % 16 == 0
    gb = al.numel() * al.element_size() / 1e9
    for kind, name in ((nat.STAT_MINMAX, 'minmax'), (nat.STAT_ABSMAX, 'absmax')):
        ms_c = t(lambda: nat.stats(kind, al, outer, ch, inner))
        ms_r = t(lambda: nat.stats(kind, mis, outer, ch, inner), n=3)
        print(f'{name} {shape} {str(dt)[6:]}: cols {ms_c*1e3:.0f} us ({gb/ms_c:.2f} TB/s)  rows {ms_r*1e3:.0f} us ({gb/ms_r:.2f} TB/s)', flush=True)
