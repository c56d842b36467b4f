"""CPU: ``bench.py --gpus N`` without a torchrun environment must start N ranks itself or fail loudly -- never
report an N-GPU number from one process."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_more_ranks_than_devices_is_a_loud_error():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(['--gpus', str(max(have + 1, 2)), '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert 'ranks requested' in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


def test_gpus_flag_spawns_one_process_per_rank():
    """with a CPU backend the device check is skipped and the self-spawn is exercised: torchrun starts 2 ranks, each of
    which stops at the engine's own 'needs a GPU' check (no GPU here) -- two processes, no JSON line"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('GPU box: covered by the real run')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--backend', 'gloo'])
    out = r.stderr + r.stdout
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout
    assert out.count('bench.py needs a ROCm GPU') >= 2, out[-2000:]


def test_world_size_mismatch_is_rejected():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout)
