"""TEST INFRASTRUCTURE (not product): one worker of bench.py's all-cores `cpu_baseline` leg.

    python oracle/cpu_baseline_worker.py <problem.npz> <budget_s>

Runs oracle/cdae_oracle.py:sparse_step (the NumPy restatement of the sampled CDAE step) over and over on the compacted problem bench.py
wrote, for `budget_s` seconds, on ONE thread, and prints "<steps> <seconds>".  bench.py starts one of these per host core — each
on a private copy of the tables — and adds the rates up: the CPU figure a data-parallel host implementation could reach, stated as
cores = processes.  Never imported by the product.
"""
import os
import sys
import time

for v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ[v] = '1'

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cdae_oracle as co  # noqa: E402


def main():
    z = np.load(sys.argv[1], allow_pickle=False)
    budget = float(sys.argv[2])
    p = {k: z['p_' + k].copy() for k in ('W', 'W_', 'V', 'b', 'b_')}
    optimizer = str(z['optimizer'])
    st = co.sparse_state(p, optimizer)
    off = z['kept_off']
    kept = [z['kept_flat'][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    cu, ci, y = z['cu'], z['ci'], z['y']
    q, lr, reg = float(z['q']), float(z['lr']), float(z['reg'])
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget:
        co.sparse_step(p, st, n, cu, ci, y, kept, q, lr, reg, 'bce', optimizer)
        n += 1
    print(n, time.perf_counter() - t0, flush=True)


if __name__ == '__main__':
    main()
