"""p2s_vanilla (or any named model) as complete shapes on the abc_minimal test shape at 256^3: queries/s for the
environment it is started in (A/B of the scheduling knobs).    python tools/vanilla_fixture.py [model] [encoder]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from points2surf_amd import engine, synth   # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else 'p2s_vanilla'
enc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pts = np.load(os.path.join(REPO, 'tests', 'golden', 'abc_minimal', '04_pts',
                           '00994122_57d9d4755722f9d2d7436f0a_trimesh_000.xyz.npy')).astype(np.float32)
w, cfg = synth.make_weights(model)
m = engine.Model(w, dict(cfg, encoder_bf16=enc))
rng = engine.Rng(40938661)
cloud = engine.Cloud(pts)
engine.infer_shape(m, cloud, rng, 256, 3, want_queries=False, q_end=20000)
torch.cuda.synchronize()
ts = []
for _ in range(2):
    t0 = time.time()
    c = engine.Cloud(pts)
    sdf, _ = engine.infer_shape(m, c, rng, 256, 3, want_queries=False)
    sdf.cpu()
    c.close()
    ts.append(time.time() - t0)
knobs = {k: v for k, v in os.environ.items() if k.startswith('P2S_')}
print('%s enc=%d %s: %.1f ms -> %.0f queries/s' % (model, enc, knobs, min(ts) * 1e3, sdf.shape[0] / min(ts)), flush=True)
