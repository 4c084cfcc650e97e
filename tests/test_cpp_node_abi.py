"""The C++ autograd nodes (brevitas_amd/csrc/bvq_autograd.cpp) take the descriptor layout and every prototype from
include/bvq.h and must refuse a libbvq.so of another ABI version instead of handing it mis-laid-out arguments: init()
against a stand-in library that exports every entry point the node resolves but reports another version."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'brevitas_amd', 'csrc', 'bvq_autograd.cpp')


def _entries():
    text = open(SRC).read()
    block = text[text.index('#define BVQ_ENTRIES(X)'):text.index('#define BVQ_DECLARE')]
    return re.findall(r'X\((bvq_\w+)\)', block)


def test_node_includes_the_header_and_hashes_it():
    text = open(SRC).read()
    assert '#include "bvq.h"' in text and 'struct QuantDesc' not in text   # no hand-kept copy of bvq_quant_desc
    assert 'BVQ_ABI_VERSION' in text
    build = open(os.path.join(ROOT, 'brevitas_amd', 'csrc', 'build.py')).read()
    assert "'include', 'bvq.h'" in build.split('def build_autograd')[1]     # a header change rebuilds the node


def test_node_refuses_a_library_of_another_abi_version(tmp_path):
    from brevitas_amd import _native as nat
    from brevitas_amd.core.quant import _fused
    mod = _fused._fast_module()
    if not mod:
        pytest.skip('brevitas_amd/_bvq_autograd.so is not built')
    assert mod.abi_version() == nat.ABI_VERSION
    names = _entries()
    assert 'bvq_abi_version' in names and len(names) > 10
    src = tmp_path / 'fake.c'
    src.write_text('\n'.join('int %s(void) { return %d; }' % (n, 999 if n == 'bvq_abi_version' else 0) for n in names))
    lib = tmp_path / 'libbvq_fake.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-o', str(lib), str(src)])
    with pytest.raises(RuntimeError, match='ABI 999'):
        mod.init(str(lib), None)
    mod.init(nat.LIB_PATH, _fused._fast_backward_fallback)   # back to the real library
