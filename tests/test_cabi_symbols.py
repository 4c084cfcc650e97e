"""The C-ABI library loads on a machine without a GPU and exports every function include/bvq.h
declares (no compute calls here); descriptor layout and argument checking are host-side."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'bvq.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bvq_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert 'bvq_fakequant_fwd' in names and 'bvq_fakequant_bwd' in names and 'bvq_stats' in names
    assert len(names) >= 16


def test_library_exports_every_declared_symbol():
    from brevitas_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_native.EXPORTS) == declared_functions()


def test_abi_version_and_error_channel():
    from brevitas_amd import _native as nat
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'bvq.h')).read()
    declared = int(re.search(r'#define BVQ_ABI_VERSION (\d+)', header).group(1))
    assert nat.lib.bvq_abi_version() == nat.ABI_VERSION == declared
    # argument validation happens before anything touches a device
    rc = nat.lib.bvq_unary(99, nat.F32, None, None, 4, None)
    assert rc < 0 and nat.last_error()
    rc = nat.lib.bvq_fakequant_fwd(None, None, None, None, None, None, None)
    assert rc == -1 and 'descriptor' in nat.last_error()
    d = nat.QuantDesc(1, 1, 16, nat.F32, nat.BF16, nat.F32, nat.F32, 0, 0, -128.0, 127.0, 0, 0, 0, 0, 0)
    assert nat.lib.bvq_fakequant_fwd(ctypes.byref(d), None, None, None, None, None, None) == -2  # f32 x cannot compute in bf16
    d = nat.QuantDesc(1, 1, 16, nat.F32, nat.F32, nat.F32, nat.F32, 0, 0, -128.0, 127.0, 7, 0, 0, 0, 0)
    assert nat.lib.bvq_fakequant_fwd(ctypes.byref(d), None, None, None, None, None, None) == -1  # bad round mode
    assert nat.lib.bvq_stats_workspace_bytes(0, nat.BF16, 256, 512, 3136) > 0
    assert nat.lib.bvq_tie_info_bytes(512) == 512 * 8
    assert nat.lib.bvq_tie_info_bytes(0) == -1


def test_descriptor_layout_matches_header():
    """bvq_quant_desc: three int64 then twelve 4-byte fields, no padding surprises"""
    from brevitas_amd import _native as nat
    assert ctypes.sizeof(nat.QuantDesc) == 3 * 8 + 14 * 4  # 13 int32/float fields + tail padding to 8
    assert nat.QuantDesc.qmin.offset == 3 * 8 + 6 * 4


def test_missing_library_is_an_import_error(tmp_path, monkeypatch):
    """the product fails loudly when the HIP extension is absent"""
    from brevitas_amd import _native
    monkeypatch.setattr(_native, 'LIB_PATH', str(tmp_path / 'libbvq.so'))
    with pytest.raises(ImportError, match='no fallback'):
        _native._load()


def test_build_script_runs_without_a_loadable_library():
    """`python -m brevitas_amd.csrc.build` and __graft_entry__.build() must work on a fresh checkout: importing
    the package on the way to the build script may not require the library the script is about to produce."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BREVITAS_AMD_LIB='/nonexistent/libbvq.so')  # any import of _native would fail loudly
    r = subprocess.run([sys.executable, '-m', 'brevitas_amd.csrc.build'], cwd=root, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith('libbvq.so')
    r = subprocess.run([sys.executable, '-c', 'import brevitas_amd'], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'There is no fallback backend' in r.stderr  # the product import still fails loudly


def _header_prototypes():
    """name -> (return type, [parameter types]) of every function declared in include/bvq.h, types as written"""
    text = open(os.path.join(ROOT, 'include', 'bvq.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?)\s*(bvq_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', text,
                         flags=re.M | re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [] if params.strip() in ('', 'void') else [re.sub(r'\s+', ' ', p.strip()) for p in params.split(',')]
        # drop the parameter name: everything up to the last identifier
        types = [re.sub(r'\s*[A-Za-z_][A-Za-z0-9_]*$', '', p).strip() if not p.endswith('*') else p for p in ps]
        protos[name] = (re.sub(r'\s+', ' ', ret.strip()), types)
    return protos


def _ctype_of(c_type, nat):
    t = c_type.replace('const ', '').strip()
    if t in ('bvq_quant_desc*', 'bvq_quant_desc *'):
        return ctypes.POINTER(nat.QuantDesc)
    if t in ('bvq_variant_desc*', 'bvq_variant_desc *'):
        return ctypes.POINTER(nat.VariantDesc)
    if t.endswith('*') or t == 'bvq_stream_t':
        return ctypes.c_void_p
    return {'int': ctypes.c_int, 'int32_t': ctypes.c_int, 'int64_t': ctypes.c_int64, 'double': ctypes.c_double,
            'float': ctypes.c_float}[t]


def test_python_binding_signatures_match_the_header():
    """every argtypes / restype list of brevitas_amd._native mirrors the prototype in include/bvq.h: same arity,
    same kind of every parameter (pointer, 32- or 64-bit integer, double) -- a drift here corrupts arguments silently"""
    from brevitas_amd import _native as nat
    protos = _header_prototypes()
    checked = 0
    for name in nat.EXPORTS:
        fn = getattr(nat.lib, name)
        if fn.argtypes is None:      # bvq_abi_version / bvq_last_error take nothing
            assert protos[name][1] == [], name
            continue
        ret, params = protos[name]
        want = [_ctype_of(p, nat) for p in params]
        assert len(fn.argtypes) == len(want), (name, len(fn.argtypes), params)
        for i, (a, b) in enumerate(zip(fn.argtypes, want)):
            assert a is b or (a in (ctypes.c_int, ctypes.c_int32) and b in (ctypes.c_int, ctypes.c_int32)), (name, i, a, b)
        assert fn.restype is {'int': ctypes.c_int, 'int64_t': ctypes.c_int64}[ret] or \
            (fn.restype in (ctypes.c_int, ctypes.c_int32) and ret == 'int'), (name, fn.restype, ret)
        checked += 1
    assert checked >= 40


def test_headline_layout_takes_the_two_launch_backward():
    """host-side coverage rules: the stats-scaled per-channel backward of the headline activation (and of conv / linear
    weights) must be served by bvq_fakequant_bwd_stats (kernel + one finishing launch), not by the general route"""
    from brevitas_amd import _native as nat
    for dt in (nat.BF16, nat.F16, nat.F32):
        for outer, ch, inner in ((256, 512, 3136), (1, 512, 4608), (1, 8192, 8192), (128, 1024, 196)):
            d = nat.QuantDesc(outer, ch, inner, dt, dt, dt, nat.F32, 1, 0, -128.0, 127.0, 0, 0, 0, 0, 0)
            assert nat.lib.bvq_fakequant_bwd_stats_workspace_bytes(ctypes.byref(d)) > 0, (dt, outer, ch, inner)
    d = nat.QuantDesc(1, 1, 1 << 20, nat.BF16, nat.BF16, nat.F32, nat.F32, 0, 0, -128.0, 127.0, 0, 0, 0, 0, 0)
    assert nat.lib.bvq_fakequant_bwd_stats_workspace_bytes(ctypes.byref(d)) == 0  # per-tensor: the general route
