"""Loader for the golden vectors generated from the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DT_CODE = {'f32': 0, 'bf16': 1, 'f16': 2}


class Case(dict):
    """meta dict + .arr(name) -> numpy array (bf16/f16 as uint16 bit patterns), .dt(name) -> code"""

    def __init__(self, meta, arrays, idx):
        super().__init__(meta)
        self._arrays = arrays
        self._idx = idx

    def has(self, name):
        return ('c%d_%s' % (self._idx, name)) in self._arrays

    def arr(self, name):
        return self._arrays['c%d_%s' % (self._idx, name)]

    def dt(self, name):
        return DT_CODE[self['dtypes'][name]]

    def f32(self, name):
        """array widened to float32 values"""
        a = self.arr(name)
        d = self['dtypes'][name]
        if d == 'f32':
            return a.astype(np.float32)
        if d == 'bf16':
            return (a.astype(np.uint32) << 16).view(np.float32)
        if d == 'f16':
            return a.view(np.float16).astype(np.float32)
        return a

    def torch(self, name, device='cpu'):
        import torch
        a = self.arr(name)
        d = self['dtypes'][name]
        if d == 'f32':
            t = torch.from_numpy(a.copy())
        elif d in ('bf16', 'f16'):
            t = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16 if d == 'bf16' else torch.float16)
        else:
            t = torch.from_numpy(a.copy())
        return t.to(device)


_cache = {}


def load(name):
    if name not in _cache:
        z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
        meta = json.loads(bytes(z['__meta__']).decode())
        arrays = {k: z[k] for k in z.files if k != '__meta__'}
        _cache[name] = [Case(m, arrays, i) for i, m in enumerate(meta)]
    return _cache[name]


def ids(cases, keys):
    return ['-'.join(str(c.get(k)) for k in keys) + '-%d' % i for i, c in enumerate(cases)]


def bits_equal(a, b):
    """bitwise equality of two oracle arrays, treating every NaN as equal to every NaN"""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype == np.float32:
        nan = np.isnan(a) & np.isnan(b)
        return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))
    if a.dtype == np.uint16:
        return True if a.size == 0 else None  # caller must say which 16-bit format: use bits_equal16
    return bool(np.all(a == b))


def bits_equal16(a, b, fmt):
    a = np.asarray(a, dtype=np.uint16)
    b = np.asarray(b, dtype=np.uint16)
    assert a.shape == b.shape
    if fmt == 'bf16':
        isnan = lambda v: (v & 0x7fff) > 0x7f80  # noqa: E731
    else:
        isnan = lambda v: (v & 0x7fff) > 0x7c00  # noqa: E731
    return bool(np.all((a == b) | (isnan(a) & isnan(b))))


def same_bits(a, b, dtname):
    if dtname == 'f32':
        return bits_equal(np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32))
    return bits_equal16(a, b, dtname)


def mismatch_report(a, b, dtname, k=5):
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    bad = np.nonzero(a != b)[0][:k]
    return 'first mismatches at %s: got %s want %s' % (bad.tolist(), a[bad].tolist(), b[bad].tolist())
