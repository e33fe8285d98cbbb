"""SURVEY 8e(iii) — how many item rows would still travel if the H hottest item rows were replicated on every rank?

One rank's batch of the row-sharded layout (B = 65 536 triples of its own users, 10M x 1M synthetic set, q = 0.2): distinct W / W2T rows
the batch touches, and how many of them are outside the H most popular items (popularity = positives per item in the training set,
known at set-up).  Traffic model per rank and step at N ranks: cold distinct rows x 4K bytes x (N-1)/N each way for the two
all-to-alls (rows out, gradient rows back), against a dense all-reduce of the 2H replicated rows (ring: 2 x 2H x 4K x (N-1)/N bytes).
Runs on CPU (torch) on a slice of the users: python scripts/hot_rows_study.py [users]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from drecpy_amd import synth                   # noqa: E402

U = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
_, N, md, mn, a = synth.SHAPES['synth-10m']
ip, idx = synth.synth_history(10_000_000, N, md, mn, a, seed=0, user_hi=U)
pop = torch.bincount(idx.long(), minlength=N)
rank_of = torch.empty(N, dtype=torch.long)
rank_of[torch.argsort(pop, descending=True)] = torch.arange(N)
g = torch.Generator(); g.manual_seed(0)
B, K, NR = 65536, 128, 8
uid = torch.randint(0, U, (B,), generator=g)
neg = torch.rand(B, generator=g) < 5 / 6
iid = torch.where(neg, torch.randint(0, N, (B,), generator=g), idx[(ip[uid] + (torch.rand(B, generator=g) * (ip[uid + 1] - ip[uid])).long())].long())
deg = ip[uid + 1] - ip[uid]
row = torch.repeat_interleave(torch.arange(B), deg)
j = torch.arange(int(deg.sum())) - (torch.cumsum(deg, 0) - deg)[row]
items = idx[ip[uid][row] + j].long()
items = items[torch.rand(len(items), generator=g) >= 0.2]
w_rows = torch.unique(items)
o_rows = torch.unique(iid)
out = {'users_in_slice': U, 'batch': B, 'touches_W': int(len(items)), 'distinct_W': int(len(w_rows)), 'distinct_W2T': int(len(o_rows)), 'by_H': {}}
row_bytes = 4 * K
for H in (0, 1024, 4096, 16384, 65536):
    cold = int((rank_of[w_rows] >= H).sum() + (rank_of[o_rows] >= H).sum())
    hot_touch = float((rank_of[items] < H).float().mean())
    a2a = cold * row_bytes * (NR - 1) / NR                      # each way
    ar = 2 * (2 * H) * row_bytes * (NR - 1) / NR                # ring all-reduce, per rank
    out['by_H'][H] = {'cold_distinct_rows': cold, 'cold_rows_per_triple': round(cold / B, 3), 'share_of_W_touches_on_hot_rows': round(hot_touch, 3),
                      'all_to_all_MB_each_way': round(a2a / 1e6, 1), 'dense_allreduce_MB': round(ar / 1e6, 1),
                      'xgmi_ms_at_450GBps': round((2 * a2a + ar) / 450e9 * 1e3, 3)}
print(json.dumps(out, indent=1))
