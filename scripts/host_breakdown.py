"""Where a reference-mode fit() step spends its host time (ml-100k / ml-1m shaped, B = 64)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch                                                     # noqa: E402
from measure_models import frame_of                              # noqa: E402
from drecpy_amd.Dataset import InteractionDataset                # noqa: E402
from drecpy_amd.Recommender import CDAE                          # noqa: E402

for shape, K in (('ml-100k', 50), ('ml-1m', 128)):
    B = 64
    ds = InteractionDataset.read_df(frame_of(shape), verbose=False)
    m = CDAE(hidden_factors=K, corruption_level=0.2, seed=10, verbose=False)
    m.fit(ds, epochs=1, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    def fit_s(n):
        t0 = time.perf_counter()
        m.fit(ds, epochs=n, batch_size=B, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    f1, f2, f3 = fit_s(1000), fit_s(6000), fit_s(6000)
    fit = (min(f2, f3) - f1) / 5000                  # steady state: set-up cancels
    print(f'{shape}: fit(1000) {f1:.3f} s, fit(6000) {f2:.3f} / {f3:.3f} s')
    n = 500
    t0 = time.perf_counter()
    batches = [m._sample_batch(B) for _ in range(n)]
    samp = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, b in enumerate(batches):
        m._do_batch(b, step=s)
    host_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n
    b = batches[0]
    t0 = time.perf_counter()
    for s in range(n):
        bt, alive = m._engine.make_batch(b.uid, keep_off=b.keep_off, keep=b.keep, q=0.2, n_touch_slots=int(b.keep_off[-1]))
    torch.cuda.synchronize()
    mk = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for s in range(n):
        m._engine.step_dense(s, bt)
    issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / n
    print(f'{shape}: fit {fit*1e3:.3f} ms/step | _sample_batch (inline, incl. corruption stream) {samp*1e3:.3f} | _do_batch issue {host_issue*1e3:.3f} '
          f'(to completion {total*1e3:.3f}) | make_batch {mk*1e3:.3f} | step_dense issue {issue*1e3:.3f} (device {dev*1e3:.3f})')
