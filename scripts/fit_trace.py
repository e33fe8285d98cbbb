"""Reference-mode fit() of N steps (for `rocprofv3 --kernel-trace -- python3 scripts/fit_trace.py ml-100k 3000`); with `--analyse
<kernel_trace.csv>` prints the mean duration of each kernel, the idle time before it and the step period."""
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == '--analyse':
    import csv
    import re
    from collections import defaultdict
    rows = []
    for r in csv.DictReader(open(sys.argv[2])):
        m = re.search(r'drx::(\w+)', r['Kernel_Name'])
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:40]))
    rows.sort()
    rows = rows[len(rows) // 2:]
    dur, gap = defaultdict(list), defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        dur[b[2]].append((b[1] - b[0]) / 1e3)
        gap[b[2]].append((b[0] - a[1]) / 1e3)
    first = [x for x in rows if x[2] == rows[0][2]]
    per = [(b[0] - a[0]) / 1e3 for a, b in zip(first, first[1:])]
    print('step period us: mean %.1f median %.1f' % (sum(per) / len(per), sorted(per)[len(per) // 2]))
    for k in dur:
        print('%-28s n=%5d  runs %.1f us, idle before it %.1f us (median %.1f)' % (k, len(dur[k]), sum(dur[k]) / len(dur[k]),
              sum(gap[k]) / len(gap[k]), sorted(gap[k])[len(gap[k]) // 2]))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import CDAE                          # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else 'ml-100k'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
m = CDAE(hidden_factors=50 if shape == 'ml-100k' else 128, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=n, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
