"""GPU: RCCL really executes the distributed code of this repository -- at world size 1, the only size a one-GPU box
offers.  ``P2S_DIST_FORCE=1`` makes sharding.init_process_group create the group also for one
rank: backend nccl (= RCCL on ROCm) initialised with ``device_id``, then exactly the calls the N-GPU runs make --
``gather_variable`` (one all_gather of the sizes + one padded gather), ``all_reduce`` (bench.py's max-over-ranks
timing), ``barrier``, the stream hand-off through the group's store, ``destroy_process_group`` -- directly, through
``bench.py --gpus 1`` and through the drop-in."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR',
                                                            'P2S_DIST_BACKEND', 'P2S_BENCH_SHARE_GPU')}
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env.update(P2S_DIST_FORCE='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    return env


SCRIPT = r'''
import numpy as np, torch, torch.distributed as dist
from points2surf_amd import sharding, engine
world, rank, local = sharding.init_process_group()
assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1, dist.get_backend()
dev = torch.device('cuda', local)
assert sharding.collective_device(dev) == dev
t = torch.arange(12345, dtype=torch.float32, device=dev) * 0.5
parts = sharding.gather_variable(t, dst=0)                 # all_gather of sizes + padded gather, on the GPU, over RCCL
assert len(parts) == 1 and parts[0].is_cuda and torch.equal(parts[0], t)
empty = sharding.gather_variable(torch.empty((0,), dtype=torch.float32, device=dev), dst=0)
assert empty[0].shape[0] == 0
x = torch.tensor([2.5], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
n = torch.tensor([7], dtype=torch.int64, device=dev)
dist.all_reduce(n, op=dist.ReduceOp.SUM)
assert float(x.item()) == 2.5 and int(n.item()) == 7
sharding.barrier()
# the stream hand-off through the group's rendezvous store, with real device generators
assert sharding.stream_handoff_enabled()
a, b = engine.Rng(40938661), engine.Rng(40938661)
cloud = engine.Cloud(np.load('tests/golden/cloud_abc_00994122.npy'))
a.subsample_uniform(cloud, 3, 1000, want_pts=False)        # advance a
blob = sharding.StreamHandoff.pack([a])
h = sharding.StreamHandoff('t', [0, 0], rank=0)
h.store.set(h._key(1), blob)
h.store.wait([h._key(1)])
sharding.StreamHandoff.unpack(h.store.get(h._key(1)), [b])
ia, _ = a.subsample_uniform(cloud, 2, 1000, want_pts=False)
ib, _ = b.subsample_uniform(cloud, 2, 1000, want_pts=False)
assert torch.equal(ia, ib)
sharding.barrier()
dist.destroy_process_group()
print('RCCL_WORLD1_OK')
'''


def test_sharding_collectives_run_over_rccl_at_world_size_one():
    r = subprocess.run([sys.executable, '-c', SCRIPT], env=_env(), cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RCCL_WORLD1_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_runs_its_distributed_path_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--res', '32', '--cpu-seconds', '0', '--no-secondary'], env=_env(), cwd=REPO, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['config']['collective'].startswith('nccl (world 1)'), d['config']
    assert d['n_gpus'] == 1 and d['value'] > 0
    g = d['self_check']['vs_reference_golden']
    assert g['sign_flips'] == 0 and g['max_abs_dsdf'] < 1e-4


def test_dropin_initialises_rccl_and_passes_its_barriers(tmp_path):
    """python -m points2surf_amd.dropin.run <script>: the drop-in's points_to_surf_eval with the RCCL group up"""
    import numpy as np
    script = tmp_path / 'driver.py'
    script.write_text('''
import os, sys
import numpy as np
import torch.distributed as dist
from source import points_to_surf_eval as ev
from points2surf_amd import synth
root, out, models = sys.argv[1:4]
os.makedirs(os.path.join(root, '04_pts'))
np.save(os.path.join(root, '04_pts', 'a.xyz.npy'), np.load(sys.argv[4]))
open(os.path.join(root, 'testset.txt'), 'w').write('a\\n')
synth.write_model_files(models, 'p2s_max')
opt = ev.parse_arguments(['--indir', root, '--outdir', out, '--dataset', 'testset.txt', '--modeldir', models, '--models',
                          'p2s_max', '--query_grid_resolution', '32', '--epsilon', '3'])
opt.reconstruction = True
ev.points_to_surf_eval(opt)
assert dist.is_initialized() and dist.get_backend() == 'nccl'
dist.destroy_process_group()
print('DROPIN_RCCL_OK')
''')
    env = _env()
    r = subprocess.run([sys.executable, '-m', 'points2surf_amd.dropin.run', str(script), str(tmp_path / 'ds'),
                        str(tmp_path / 'out'), str(tmp_path / 'models'),
                        os.path.join(REPO, 'tests', 'golden', 'cloud_abc_00994122.npy')],
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'DROPIN_RCCL_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    sdf = np.load(str(tmp_path / 'out' / 'rec' / 'dist_ms' / 'a.xyz.npy'))
    ref = np.load(os.path.join(REPO, 'tests', 'golden', 'ref_p2s_max_grid32.npz'))['sdf_full']
    assert sdf.shape == ref.shape and np.abs(sdf - ref).max() < 1e-5
