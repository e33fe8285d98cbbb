"""L2 hit rate per drx kernel from a `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` pass (MI355X_MICROARCH.md, L2 section:
hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)); median over launches.
Usage: python profiles/pmc_l2.py <counter_collection.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def main(path, out):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'drx::' in k:
            acc[k.split('<')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
    res = {}
    for k, c in sorted(acc.items()):
        h, m = med(c.get('TCC_HIT_sum', [])), med(c.get('TCC_MISS_sum', []))
        res[k] = {'TCC_HIT_sum': h, 'TCC_MISS_sum': m, 'l2_hit_rate': (h / (h + m)) if h + m else None}
        print(f"{k:40s} hit {h:14.0f} miss {m:14.0f} rate {res[k]['l2_hit_rate']}")
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
