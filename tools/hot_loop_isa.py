#!/usr/bin/env python
"""ISA of a kernel's hot loop, with the instructions counted per class and per 16-byte chunk.

    python tools/hot_loop_isa.py [--part bwd_bf16] [--symbol MANGLED] [--chunks-per-trip N] [--asm FILE]

Compiles brevitas_amd/csrc/bvq_fakequant_<part>.hip to gfx950 assembly (hipcc -S --cuda-device-only, the
library's own flags; about a minute for a backward family), takes one kernel's body, finds its innermost
loops (a conditional branch back to an earlier label) and prints the one holding the most global
loads / stores: the listing, the instruction mix and the count per 16-byte chunk a lane moves.
The default symbol is the headline step's backward (bf16 tensor and arithmetic, 8 elements per lane,
round-half-even, scale gradient with the arg-max search of the statistic, non-temporal policy).
"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, '.')
from brevitas_amd.csrc.build import CSRC, FLAGS, ROOT, _hipcc  # noqa: E402

HEADLINE_BWD = '_ZN3bvq20fakequant_bwd_kernelIDF16bDF16bLi8ELi0ELi5ELb1ELb1EEEvNS_9QuantArgsE'  # one-launch form (mode 5)


def classify(op):
    if op.startswith(('global_load', 'buffer_load', 'flat_load')):
        return 'vector memory load'
    if op.startswith(('global_store', 'buffer_store', 'flat_store')):
        return 'vector memory store'
    if op.startswith(('global_atomic', 'buffer_atomic', 'flat_atomic')):
        return 'vector memory atomic'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith('s_waitcnt'):
        return 's_waitcnt'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'scalar memory'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('v_pk_'):
        return 'VALU packed (2 lanes of work per issue)'
    if op.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_exp', 'v_log', 'v_div_')):
        return 'VALU transcendental / division helpers'
    if op.startswith('v_cmp') or op.startswith('v_cmpx'):
        return 'VALU compare'
    if op.startswith('v_cndmask'):
        return 'VALU select'
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_mov', 'v_accvgpr')):
        return 'VALU move'
    if op.startswith('v_'):
        return 'VALU arithmetic'
    return 'other'


def kernel_body(asm, symbol):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol + ':'))
    # (the function's end marker, not its first s_endpgm: the compiler may lay blocks out behind an early exit)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    return lines[start:end]


def loops(body):
    """(first line, last line) of every backward conditional branch's span"""
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB[0-9_]+):', l)
        if m:
            label_at[m.group(1)] = i
    spans = []
    for i, l in enumerate(body):
        m = re.match(r'\s+s_c?branch\S*\s+(\.LBB[0-9_]+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            spans.append((label_at[m.group(1)], i))
    return spans


def instructions(body, span):
    out = []
    for l in body[span[0]:span[1] + 1]:
        t = l.split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            continue
        out.append(t)
    return out


def main():
    args = sys.argv[1:]

    def opt(name, default):
        for i, a in enumerate(args):
            if a == name:
                return args[i + 1]
        return default
    part, symbol = opt('--part', 'bwd_bf16'), opt('--symbol', HEADLINE_BWD)
    per_trip = int(opt('--chunks-per-trip', '0'))
    asm_path = opt('--asm', None)
    if asm_path is None:
        asm_path = '/tmp/bvq_fakequant_%s.s' % part
        cmd = [_hipcc()] + [f for f in FLAGS if not f.startswith('-W')] + \
            ['-S', '--cuda-device-only', '-I', os.path.join(ROOT, 'include'),
             os.path.join(CSRC, 'bvq_fakequant_%s.hip' % part), '-o', asm_path]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
    body = kernel_body(open(asm_path).read(), symbol)
    name = subprocess.run(['c++filt', symbol], capture_output=True, text=True).stdout.strip()
    meta = {}
    asm = open(asm_path).read()
    m = re.search(r'\.amdhsa_kernel ' + re.escape(symbol) + r'(.*?)\.end_amdhsa_kernel', asm, re.S)
    if m:
        for key in ('next_free_vgpr', 'next_free_sgpr', 'accum_offset'):
            mm = re.search(r'\.amdhsa_' + key + r'\s+(\S+)', m.group(1))
            if mm:
                meta[key] = mm.group(1)
    spans = loops(body)
    # innermost loops only: spans that contain no other span
    inner = [s for s in spans if not any(o != s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]

    def mem_ops(s):
        return sum(1 for t in instructions(body, s) if classify(t.split()[0]).startswith('vector memory'))

    # The kernel holds one copy of the walk per (zero-point is +0, fused ReLU, division form) combination, chosen by
    # wave-uniform branches.  The headline step runs the copy with the reciprocal instead of the IEEE division, a
    # +0 zero-point and no ReLU: among the loops with the most memory instructions, the one without v_div_* and
    # with the fewest instructions.
    most = max(mem_ops(s) for s in inner)
    cands = [s for s in inner if mem_ops(s) == most]
    nodiv = [s for s in cands if not any(t.startswith('v_div_') for t in instructions(body, s))]
    hot = min(nodiv or cands, key=lambda s: len(instructions(body, s)))
    ins = instructions(body, hot)
    mix = collections.Counter(classify(t.split()[0]) for t in ins)
    loads = [t for t in ins if classify(t.split()[0]) == 'vector memory load']
    stores = [t for t in ins if classify(t.split()[0]) == 'vector memory store']
    print('kernel:', name)
    print('registers:', ', '.join('%s %s' % kv for kv in meta.items()))
    print('whole kernel: %d instructions, %d loops (%d innermost, %d of them copies of the walk)' %
          (len(instructions(body, (0, len(body) - 1))), len(spans), len(inner), len(cands)))
    print('hot loop: %d instructions per trip; %d vector loads (%s), %d vector stores (%s)' %
          (len(ins), len(loads), ', '.join(sorted(set(t.split()[0] for t in loads))),
           len(stores), ', '.join(sorted(set(t.split()[0] for t in stores)))))
    chunks = per_trip or len(stores) or 1
    print('16-byte chunks of dx written per lane per trip: %d' % chunks)
    waits = [t for t in ins if t.startswith('s_waitcnt')]
    print('waits in the loop: ' + '; '.join(w.replace('s_waitcnt ', '') for w in waits))
    print()
    print('%-46s %6s %10s' % ('class', 'count', 'per chunk'))
    for k, v in sorted(mix.items(), key=lambda kv: -kv[1]):
        print('%-46s %6d %10.1f' % (k, v, v / chunks))
    valu = sum(v for k, v in mix.items() if k.startswith('VALU'))
    print('%-46s %6d %10.1f' % ('VALU, all classes', valu, valu / chunks))
    print('%-46s %6d %10.1f' % ('all', len(ins), len(ins) / chunks))
    print()
    print('listing (hot loop):')
    for l in body[hot[0]:hot[1] + 1]:
        t = l.split(';')[0].rstrip()
        if t.strip():
            print(t)


if __name__ == '__main__':
    main()
