"""Experiment: does the placement of the two draw workers relative to the main thread explain the bimodal fit() time?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from measure_models import frame_of
from drecpy_amd.Dataset import InteractionDataset
from drecpy_amd.Recommender import CDAE
from concurrent.futures import ThreadPoolExecutor

def sib(c):
    try:
        return open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list').read().strip()
    except Exception as e:
        return str(e)
print('siblings of cpu8:', sib(8), '| cpu16:', sib(16), '| stat cpu field:', open('/proc/self/stat').read().split()[38])
ds = InteractionDataset.read_df(frame_of('ml-100k'), verbose=False)
mode = sys.argv[1]
m = CDAE(hidden_factors=50, corruption_level=0.2, seed=10, verbose=False)
m.fit(ds, epochs=10, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
if mode != 'free':
    s8 = [int(x) for x in sib(8).replace('-', ',').split(',')]
    main_cpu = 8
    wcpus = {'apart': [16, 24], 'siblings': [s8[-1], s8[-1]], 'same': [8, 8]}[mode]
    os.sched_setaffinity(0, {main_cpu})
    def init(c):
        os.sched_setaffinity(threading.get_native_id(), {c})
    m._draw_pools = [ThreadPoolExecutor(max_workers=1, initializer=init, initargs=(wcpus[g],)) for g in range(2)]
for rep in range(6):
    t0 = time.perf_counter()
    m.fit(ds, epochs=5000, batch_size=64, learning_rate=1e-3, reg_rate=1e-3, neg_ratio=5)
    torch.cuda.synchronize()
    print(mode, f'fit(5000) {time.perf_counter() - t0:.3f} s')
