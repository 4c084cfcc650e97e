"""Build libbvq.so (the C-ABI HIP library) in-tree for gfx950.

    python -m brevitas_amd.csrc.build [--force]

hipcc cross-compiles without a GPU; the resulting brevitas_amd/libbvq.so is git-ignored but travels
to the GPU box with the source snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(CSRC)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'libbvq.so')
OBJ_DIR = os.path.join(ROOT, 'build', 'bvq')

# (source, extra -D flags, object name).  The quantizer is split by kernel family: forward; backward entry points with the
# column-mapped / finishing kernels; the row-mapped backward kernel per dtype family (the long poles of the build).
SOURCES = [('bvq_common.hip', [], 'bvq_common.o'), ('bvq_elementwise.hip', [], 'bvq_elementwise.o'),
           ('bvq_stats.hip', [], 'bvq_stats.o'), ('bvq_select.hip', [], 'bvq_select.o'),
           ('bvq_variants.hip', [], 'bvq_variants.o'), ('bvq_fakequant_fwd.hip', [], 'bvq_fakequant_fwd.o'),
           ('bvq_fakequant_bwd.hip', [], 'bvq_fakequant_bwd.o'),
           ('bvq_fakequant_bwd_bf16.hip', [], 'bvq_fakequant_bwd_bf16.o'),
           ('bvq_fakequant_bwd_f16.hip', [], 'bvq_fakequant_bwd_f16.o'),
           ('bvq_fakequant_bwd_f32.hip', [], 'bvq_fakequant_bwd_f32.o')]
HEADERS = ['bvq_common.h', 'bvq_quant_math.h', 'bvq_ties.h', 'bvq_sums.h', 'bvq_fakequant.h', 'bvq_fakequant_bwd.h',
           os.path.join(ROOT, 'include', 'bvq.h')]

# -ffp-contract=off: the reference rounds after every op; a contracted mul+add would not.
# hipcc's default fp32 division is correctly rounded (no -ffast-math, no approximate reciprocal).
# --offload-compress: the gfx950 code objects are stored compressed in the fat binary (12.9 -> ~4 MB); the HIP runtime
# unpacks them when the library's first kernel is launched.
FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '--offload-compress',
    '-Wall', '-Wno-unused-function', '-Wno-sometimes-uninitialized', '-Wno-uninitialized',
    '-Wno-unused-variable']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: brevitas_amd needs the ROCm toolchain to build libbvq.so')


def _digest():
    h = hashlib.sha256()
    for f in sorted(set(src for src, _, _ in SOURCES)) + HEADERS + [os.path.abspath(__file__)]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        with open(path, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def source_digest():
    """digest of the sources + flags brevitas_amd/libbvq.so was built from, or None when the library on disk is not the
    build of the sources on disk (bench.py stamps its tracked PMC traffic figures with it)"""
    stamp = os.path.join(OBJ_DIR, 'stamp')
    dig = _digest()
    if os.path.exists(LIB) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == dig:
                return dig
    return None


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: developer experiments only (tools/microbench.py --lib): extra -D flags, other output"""
    lib_path = out or LIB
    obj_dir = OBJ_DIR if not out else os.path.join(OBJ_DIR, os.path.basename(out).replace('.', '_'))
    stamp = os.path.join(obj_dir, 'stamp')
    dig = (_digest() + ' ' + ' '.join(defines)).strip()
    if not force and os.path.exists(lib_path) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == dig:
                return lib_path
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(item):
        src, part_defines, objname = item
        obj = os.path.join(obj_dir, objname)
        cmd = [hipcc] + FLAGS + ['-D' + d for d in list(defines) + part_defines] + \
            ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path + '.tmp'] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    os.replace(lib_path + '.tmp', lib_path)
    with open(stamp, 'w') as fh:
        fh.write(dig)
    return lib_path


AUTOGRAD_SO = os.path.join(PKG, '_bvq_autograd.so')


def build_autograd(force=False, verbose=False):
    """Compile the optional C++ autograd node (bvq_autograd.cpp: host code only, g++ against torch's headers) in-tree ->
    brevitas_amd/_bvq_autograd.so.  The package works without it (Python Function); returns the path or None."""
    src = os.path.join(CSRC, 'bvq_autograd.cpp')
    import torch
    h = hashlib.sha256()
    for path in (src, os.path.join(ROOT, 'include', 'bvq.h')):   # the node takes every prototype from the header
        with open(path, 'rb') as fh:
            h.update(fh.read())
    h.update(torch.__version__.encode())
    stamp = os.path.join(ROOT, 'build', 'autograd', 'stamp')
    if not force and os.path.exists(AUTOGRAD_SO) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == h.hexdigest():
                return AUTOGRAD_SO
    from torch.utils import cpp_extension
    bdir = os.path.join(ROOT, 'build', 'autograd')
    os.makedirs(bdir, exist_ok=True)
    cpp_extension.load(name='_bvq_autograd', sources=[src], build_directory=bdir, extra_cflags=['-O2', '-std=c++17', '-D__HIP_PLATFORM_AMD__=1'],
                       extra_include_paths=[os.path.join(ROOT, 'include'), os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'include')],
                       extra_ldflags=['-ldl'], verbose=verbose, is_python_module=False)
    built = os.path.join(bdir, '_bvq_autograd.so')
    shutil.copyfile(built, AUTOGRAD_SO)
    with open(stamp, 'w') as fh:
        fh.write(h.hexdigest())
    return AUTOGRAD_SO


if __name__ == '__main__':
    defs = [a[2:] for a in sys.argv[1:] if a.startswith('-D')]
    outs = [a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--out=')]
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv, defines=defs,
                out=os.path.abspath(outs[0]) if outs else None))
