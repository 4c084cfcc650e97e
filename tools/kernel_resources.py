#!/usr/bin/env python
"""Register / occupancy table of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py brevitas_amd/csrc/bvq_fakequant_bwd_bf16.hip [-DFLAG ...] [--filter=NAME]
"""
import re
import subprocess
import sys

sys.path.insert(0, '.')
from brevitas_amd.csrc.build import FLAGS, _hipcc  # noqa: E402


def main():
    src = sys.argv[1]
    extra = [a for a in sys.argv[2:] if a.startswith('-D')]
    filt = [a.split('=', 1)[1] for a in sys.argv[2:] if a.startswith('--filter=')]
    cmd = [_hipcc()] + FLAGS + extra + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    demangle = {}
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r'remark:\s+(.*)', line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith('Function Name:'):
            cur = {'name': t.split(':', 1)[1].strip().split(' ')[0]}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip().split(' ')[0]
    names = [r_['name'] for r_ in rows]
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
    for r_, d in zip(rows, out):
        r_['name'] = d
    print('%-6s %-6s %-5s %-7s %s' % ('VGPR', 'SGPR', 'occ', 'scratch', 'kernel'))
    for r_ in rows:
        if filt and not any(f in r_['name'] for f in filt):
            continue
        print('%-6s %-6s %-5s %-7s %s' % (r_.get('VGPRs'), r_.get('TotalSGPRs'), r_.get('Occupancy [waves/SIMD]'),
                                         r_.get('ScratchSize [bytes/lane]'), r_['name'][:150]))


if __name__ == '__main__':
    main()
