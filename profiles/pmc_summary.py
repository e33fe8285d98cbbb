"""Summarises rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes) into per-launch HBM traffic of the drx kernels.

Units/corrections from the guide: both counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a
wide coalesced (16 B/lane) streaming read -> doubled here (all drx row accesses are 16 B/lane); WRITE_SIZE is exact for
16-B-per-lane stores.  Usage: python profiles/pmc_summary.py <fetch counter_collection.csv> <write ...csv> <out.json>
"""
import csv
import json
import sys
from collections import defaultdict


def base_name(kernel):
    """'void drx::k_seg_reduce<32, 1, drx::DirectPolicy>(drx::SegBufs, ...)' -> 'drx::k_seg_reduce' (template arguments and
    the return type differ between rocprofv3 versions; the bench looks kernels up by this name)."""
    import re
    m = re.search(r'drx::(\w+)', kernel)
    return 'drx::' + m.group(1) if m else kernel.split('(')[0]


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[base_name(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return acc


def main(fetch_csv, write_csv, out):
    f = per_kernel(fetch_csv, 'FETCH_SIZE')
    w = per_kernel(write_csv, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        if 'drx::' not in k and 'rocprim' not in k:
            continue
        fv, wv = f.get(k, []), w.get(k, [])
        # the first dispatches are warm-up steps: use the median
        med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
        res[k] = {'launches': max(len(fv), len(wv)), 'fetch_bytes_corrected': 2.0 * med(fv) * 1024.0,
                  'write_bytes': med(wv) * 1024.0, 'fetch_raw_kib': med(fv), 'write_raw_kib': med(wv)}
        res[k]['hbm_bytes_per_launch'] = res[k]['fetch_bytes_corrected'] + res[k]['write_bytes']
    json.dump(res, open(out, 'w'), indent=1)
    for k, v in res.items():
        print(f"{k[:70]:70s} n={v['launches']:4d} fetch {v['fetch_bytes_corrected'] / 1e6:9.1f} MB  write {v['write_bytes'] / 1e6:9.1f} MB")


if __name__ == '__main__':
    main(*sys.argv[1:4])
