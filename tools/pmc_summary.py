"""Per-kernel means of the hardware counters collected by tools/pmc_passes.sh (rocprofv3 --pmc, one pass per group)."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profile import short  # noqa: E402


def main():
    root = sys.argv[1]
    table = defaultdict(dict)   # kernel -> counter -> mean per dispatch
    durs = defaultdict(list)
    for cc in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(cc) as fh:
            for row in csv.DictReader(fh):
                acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
                if 'Start_Timestamp' in row and row.get('End_Timestamp'):
                    durs[short(row['Kernel_Name'])].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        for k, cs in acc.items():
            for c, v in cs.items():
                table[k][c] = sum(v) / len(v)
    for kt in sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)):
        with open(kt) as fh:
            for row in csv.DictReader(fh):
                durs[short(row['Kernel_Name'])].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    print('# rocprofv3 --pmc passes (tools/pmc_passes.sh): mean per dispatch\n')
    for k in sorted(table, key=lambda k: -sum(durs.get(k, [0])) ):
        d = durs.get(k)
        print('## %s%s\n' % (k, '  (avg %.1f us under the counter passes, %d dispatches)' % (sum(d) / len(d), len(d)) if d else ''))
        print('| counter | mean per dispatch |')
        print('|---|---|')
        for c in sorted(table[k]):
            print('| %s | %.6g |' % (c, table[k][c]))
        print()


if __name__ == '__main__':
    main()
