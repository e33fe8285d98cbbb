"""Median per-dispatch value of every counter of every drx kernel in rocprofv3 --pmc counter_collection CSVs.
Usage: python profiles/pmc_generic.py <out.json> <csv> [<csv> ...]"""
import csv
import json
import re
import sys
from collections import defaultdict


def main(out, *paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            m = re.search(r'drx::(\w+)', r['Kernel_Name'])
            if m:
                acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} | {'dispatches': max(len(v) for v in cs.values())} for k, cs in acc.items()}
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    for k, cs in sorted(res.items()):
        print(k, {c: (round(v) if v > 100 else round(v, 3)) for c, v in cs.items()})


if __name__ == '__main__':
    main(*sys.argv[1:])
