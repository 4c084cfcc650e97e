"""Condense rocprofv3 CSV output (tools/profile.sh) into the summary committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    hits = glob.glob(os.path.join(root, '**', pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace('void ', '')
    for key in ('fakequant_bwd_kernel', 'fakequant_fwd_kernel', 'absmax_kernel', 'minmax_kernel', 'stat_finish_kernel',
                'channel_sum_kernel', 'tie_apply_first_kernel', 'tie_apply_list_kernel', 'tie_apply_full_kernel',
                'tie_init_kernel', 'tie_scan_kernel', 'running_stats_kernel', 'map_kernel'):
        if key in name:
            return 'bvq::' + key
    return name[:70]


def main():
    root = sys.argv[1]
    out = []
    kt = find(os.path.join(root, 'kt'), '*kernel_trace.csv')
    per = defaultdict(list)
    if kt:
        with open(kt) as fh:
            for row in csv.DictReader(fh):
                per[short(row['Kernel_Name'])].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        total = sum(sum(v) for v in per.values())
        out.append('## rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 50\n')
        bj = os.path.join(root, 'kt_bench.json')
        if os.path.exists(bj):
            out.append('bench line under the profiler: `%s`\n' % open(bj).read().strip()[:600])
        out.append('| kernel | calls | avg us | min us | max us | total ms | % |')
        out.append('|---|---|---|---|---|---|---|')
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            out.append('| %s | %d | %.1f | %.1f | %.1f | %.3f | %.1f |' % (
                k, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3, 100 * sum(v) / total))
    traffic = {}
    for tag, counter in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
        cc = find(os.path.join(root, tag), '*counter_collection.csv')
        if not cc:
            continue
        acc = defaultdict(list)
        with open(cc) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    acc[short(row['Kernel_Name'])].append(float(row['Counter_Value']))
        out.append('\n## rocprofv3 --pmc %s (own pass), per dispatch\n' % counter)
        out.append('| kernel | dispatches | mean %s (KiB as reported) |' % counter)
        out.append('|---|---|---|')
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            out.append('| %s | %d | %.1f |' % (k, len(v), sum(v) / len(v)))
            traffic.setdefault(k, {})[counter] = sum(v) / len(v)
    print('\n'.join(out))
    with open(os.path.join(root, 'traffic_raw.json'), 'w') as fh:
        json.dump(traffic, fh, indent=1)


if __name__ == '__main__':
    main()
