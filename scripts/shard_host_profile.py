"""Host-side issue time vs device time of the row-sharded step at world 1 (where is the sharded path host-bound?)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drecpy_amd import synth
from drecpy_amd.dist import ShardedCdae
U, N = 1_000_000, 1_000_000
_, _, md, mn, a = synth.SHAPES['synth-10m']
ip, idx = synth.synth_history(U, N, md, mn, a, seed=0, device='cuda')
m = ShardedCdae(U, N, 128, 0, 1, 'cuda:0', ip, idx, q=0.2)
eng = m.engine
B = 65536
uid, iid, y, ko = eng.sample_device(B, 5, 1, n_items=N)
torch.cuda.synchronize()
bt, alive = eng.make_batch(uid, iid, y, keep_off=ko, q=0.2, mask_seed=1, n_touch_slots=int(ko[-1].item()))
ops = m.ops
for _ in range(3):
    m.step(0, bt)
torch.cuda.synchronize()
def stage(name, fn, acc):
    t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    acc.setdefault(name, [0.0, 0.0]); acc[name][0] += t1 - t0; acc[name][1] += t2 - t0
    return r
acc = {}
for it in range(10):
    opt = ops.optim(it)
    keys, vals, bpos = stage('touches', lambda: ops.touches(bt), acc)
    idx_ = stage('index(+bounds sync)', lambda: ops.index(keys, vals), acc)
    q_item = idx_['bounds'][1]
    req = idx_['uniq_keys'][:q_item]
    rows, b2v = stage('gather_rows', lambda: ops.gather_rows(req), acc)
    stage('fwd_bwd', lambda: ops.fwd_bwd(bt, idx_['slot_of_pos'], rows, b2v, B, 0), acc)
    gc, gb2c = stage('reduce', lambda: ops.reduce(idx_, bpos, q_item, B, 0.2, opt), acc)
    stage('apply', lambda: ops.apply(req, gc, gb2c, B, opt), acc)
    gb = stage('bias_grad', lambda: ops.bias_grad(B), acc)
    stage('bias_apply', lambda: ops.bias_apply(gb, B, opt), acc)
for k, (h, d) in acc.items():
    print(f'{k:22s} host issue {h / 10 * 1e3:7.3f} ms   issue+device {d / 10 * 1e3:7.3f} ms')
