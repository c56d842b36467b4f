"""Timing of the sign-propagation row (f-1) at 256^3 on a synthetic shape (development aid)."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from points2surf_amd import engine, synth

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pts = synth.make_cloud(50000, seed=1000)
cloud = engine.Cloud(pts)
q = cloud.query_grid(res, 3)
# analytic-ish SDF: signed distance to the cloud's star-shaped blob approximated by radius comparison + noise
qn = q.cpu().numpy()
r = np.linalg.norm(qn, axis=1)
d = (0.33 - r).astype(np.float32) + (0.003 * np.random.default_rng(0).standard_normal(r.shape)).astype(np.float32)
dd = torch.from_numpy(d).cuda()
vol, it = engine.sdf_volume(q, dd, res, 5, 13.0)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    vol, it = engine.sdf_volume(q, dd, res, 5, 13.0)
torch.cuda.synchronize()
dt = (time.time() - t0) / 3
nv = res ** 3
v, f, inv = engine.marching_cubes(vol)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    v, f, inv = engine.marching_cubes(vol)
torch.cuda.synchronize()
dt_mc = (time.time() - t0) / 3
print(json.dumps({'res': res, 'marching_cubes_ms': dt_mc * 1e3, 'vertices': int(v.shape[0]), 'faces': int(f.shape[0]),
                  'GBps_volume_read_4B_per_voxel': 4.0 * nv / dt_mc / 1e9}))
print(json.dumps({'res': res, 'queries': int(q.shape[0]), 'sweeps': it, 'ms': dt * 1e3, 'ms_per_sweep': dt * 1e3 / max(it, 1),
                  'GBps_algorithmic_2B_per_voxel_sweep': 2.0 * nv * it / dt / 1e9,
                  'unknown_left': int((vol == 0).sum().item())}))
