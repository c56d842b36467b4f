import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from points2surf_amd import engine, synth
w, cfg = synth.make_weights('p2s_max')
m = engine.Model(w, cfg)
cloud = engine.Cloud(synth.make_cloud(50000, seed=1000))
rng = engine.Rng(1)
engine.infer_shape(m, cloud, rng, 256, 3, q_begin=0, q_end=4096, want_queries=False)
for nq in (1, 4096, 8192, 16384, 40960):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        engine.infer_shape(m, cloud, rng, 256, 3, q_begin=0, q_end=nq, want_queries=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    print('nq=%6d  %.2f ms  (%.2f ms per 4096-chunk)' % (nq, dt * 1e3, dt * 1e3 / max(nq / 4096, 1)))
