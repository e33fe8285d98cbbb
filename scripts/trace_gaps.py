"""Gaps between the training kernels of consecutive steps from a rocprofv3 --kernel-trace CSV (which kernels overlap the step, how
long the chip waits between dependent launches).  Usage: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'drx::(\w+)', r['Kernel_Name'])
    name = m.group(1) if m else r['Kernel_Name'][:40]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
main = [x for x in rows if x[2] in ('k_sampled_fwd_bwd', 'k_seg_reduce', 'k_sparse_tail_a', 'k_sparse_tail_b')]
main = main[len(main) // 3:]                      # skip warm-up
gap = defaultdict(list)
dur = defaultdict(list)
for a, b in zip(main, main[1:]):
    gap[a[2] + ' -> ' + b[2]].append((b[0] - a[1]) / 1e3)
for x in main:
    dur[x[2]].append((x[1] - x[0]) / 1e3)
steps = [x for x in main if x[2] == 'k_sampled_fwd_bwd']
per = [(b[0] - a[0]) / 1e3 for a, b in zip(steps, steps[1:])]
print('step period us: mean %.1f min %.1f max %.1f (n=%d)' % (sum(per) / len(per), min(per), max(per), len(per)))
for k, v in dur.items():
    print('dur  %-20s mean %.1f us' % (k, sum(v) / len(v)))
for k, v in gap.items():
    print('gap  %-45s mean %.1f us  max %.1f' % (k, sum(v) / len(v), max(v)))
side = defaultdict(list)
for x in rows[len(rows) // 3:]:
    if x[2] not in dur:
        side[x[2]].append((x[1] - x[0]) / 1e3)
for k, v in sorted(side.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print('side %-40s n=%d mean %.1f us total/step %.1f' % (k, len(v), sum(v) / len(v), sum(v) / max(len(steps), 1)))
