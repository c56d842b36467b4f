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


def test_launcher_uses_all_visible_devices_by_default(tmp_path, monkeypatch):
    """python -m points2surf_amd.dropin.run <script> without a torchrun environment: one rank per visible device, as the
    reference's DataParallel without device_ids uses all of them (source/points_to_surf_eval.py:168); under torchrun,
    with one device, or with P2S_GPUS=1 it stays in the process; more ranks than devices is an error"""
    import pytest
    from points2surf_amd.dropin import run
    assert run.ranks_to_spawn({}, 8) == 8
    assert run.ranks_to_spawn({'P2S_GPUS': '4'}, 8) == 4
    assert run.ranks_to_spawn({'P2S_GPUS': '1'}, 8) == 0
    assert run.ranks_to_spawn({'WORLD_SIZE': '8', 'RANK': '3'}, 8) == 0        # already a rank
    assert run.ranks_to_spawn({}, 1) == 0 and run.ranks_to_spawn({}, 0) == 0
    with pytest.raises(SystemExit):
        run.ranks_to_spawn({'P2S_GPUS': '16'}, 8)
    # only rank-aware scripts are ever started as several ranks -- full_run.py calls the reference's own
    # training, which N ranks would run N times into the same model files
    assert run.ranks_to_spawn({}, 8, script='/x/full_eval.py') == 8
    assert run.ranks_to_spawn({}, 8, script='/x/full_run.py') == 0 and run.ranks_to_spawn({}, 8, script='make_dataset.py') == 0
    assert run.ranks_to_spawn({'P2S_GPUS': '1'}, 8, script='/x/full_run.py') == 0
    with pytest.raises(SystemExit, match='not rank-aware'):
        run.ranks_to_spawn({'P2S_GPUS': '4'}, 8, script='/x/full_run.py')
    cmd = run.spawn_command(8, ['/x/full_eval.py', '--indir', 'd'], 4711)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node' in cmd and '8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '4711'
    assert cmd[-5:] == ['-m', 'points2surf_amd.dropin.run', '/x/full_eval.py', '--indir', 'd']
    # end to end: two ranks (P2S_GPUS=2 with a faked device count) run the SCRIPT, each with its own RANK
    script = tmp_path / 's.py'
    script.write_text("import os\nopen(os.path.join(%r, 'rank_' + os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'])\n" % str(tmp_path))
    monkeypatch.setattr(run, 'ranks_to_spawn', lambda environ=None, device_count=None, script=None: 0 if 'WORLD_SIZE' in os.environ else 2)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        monkeypatch.delenv(k, raising=False)
    assert run.main([str(script)]) == 0
    assert (tmp_path / 'rank_0').read_text() == '2' and (tmp_path / 'rank_1').read_text() == '2'
    # torchrun --nproc-per-node N -m points2surf_amd.dropin.run full_run.py -- N trainings into the same files: refused
    other = tmp_path / 'full_run.py'
    other.write_text('raise SystemExit("must not run")\n')
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(SystemExit, match='not rank-aware'):
        run.main([str(other)])
    monkeypatch.setenv('WORLD_SIZE', '1')
    with pytest.raises(SystemExit, match='must not run'):          # one process: runs
        run.main([str(other)])


def test_clock_sampler_reads_hwmon_files(tmp_path):
    """bench.ClockSampler: mean / min shader clock and power from the hwmon files while a region runs; None where there are none"""
    import time
    import bench
    hw = tmp_path / 'hwmon3'
    hw.mkdir()
    (hw / 'freq1_input').write_text('1550000000\n')
    (hw / 'power1_input').write_text('1000000000\n')
    c = bench.ClockSampler(hwmon=str(hw))
    c.start()
    time.sleep(0.1)
    (hw / 'freq1_input').write_text('1450000000\n')
    time.sleep(0.1)
    r = c.stop()
    assert r['samples'] >= 4 and r['sclk_MHz_min'] == 1450.0 and 1450.0 < r['sclk_MHz_mean'] < 1550.0
    assert abs(r['power_W_mean'] - 1000.0) < 1e-6
    none = bench.ClockSampler(hwmon=str(tmp_path / 'missing'))
    none.start()
    assert none.stop() is None
