"""raw per-kernel PMC summary (pmc_summary.py) -> the file bench.py quotes `roofline.traffic` from, keyed by the exact workload
and by the hash of the kernel sources it was measured on (bench.py:kernel_source_hash): a profile of older kernels is reported as
stale instead of being divided by fresh timings.  Usage: python profiles/pmc_make_traffic.py <raw.json> <out.json> <round tag> [workload batch optimizer l2_hit_rate.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(raw, out, tag, workload='synth-10m', batch=65536, optimizer='adagrad', l2=None):
    import bench
    kernels = json.load(open(raw))
    if l2 and os.path.exists(l2):            # the TCC_HIT / TCC_MISS pass of the same script (profiles/pmc_l2.py): the kernels' L2 hit rates
        for k, v in json.load(open(l2)).items():
            if k in kernels:
                kernels[k]['l2_hit_rate'] = v.get('l2_hit_rate')
    meta = {'workload': workload, 'batch_per_gpu': int(batch), 'n_gpus': 1, 'optimizer': optimizer, 'round': tag,
            'kernel_source_hash': bench.kernel_source_hash(),
            'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_round.sh) on `bench.py --steps 6 --warmup 2`; '
                      'FETCH_SIZE doubled (gfx950 16 B/lane correction), KiB -> bytes; median over launches; see profiles/pmc_summary.py; l2_hit_rate: a third pass, --pmc TCC_HIT_sum TCC_MISS_sum (profiles/pmc_l2.py)'}
    json.dump({'_meta': meta, 'kernels': kernels}, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:])
