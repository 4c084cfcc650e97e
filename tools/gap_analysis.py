"""Where is the GPU idle inside a headline step?  Reads a rocprofv3 kernel-trace CSV of bench.py and prints, for the
steady-state steps, each kernel's mean duration and the mean idle gap before it."""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
rows.sort()
# the steady state: the last 60 % of the launches
rows = rows[int(len(rows) * 0.4):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
order = OrderedDict()
prev_end = None
for st, en, name in rows:
    order.setdefault(name, None)
    dur[name] += en - st
    if prev_end is not None:
        gap[name] += max(0, st - prev_end)
    cnt[name] += 1
    prev_end = en
tot_d = tot_g = 0.0
n_steps = max(cnt.values())
for name in order:
    print('%-62s n=%5d  dur %8.2f us  idle-before %6.2f us' % (name, cnt[name], dur[name] / cnt[name] / 1e3, gap[name] / cnt[name] / 1e3))
    tot_d += dur[name] / n_steps / 1e3
    tot_g += gap[name] / n_steps / 1e3
print('per step: kernels %.1f us + idle %.1f us = %.1f us' % (tot_d, tot_g, tot_d + tot_g))
