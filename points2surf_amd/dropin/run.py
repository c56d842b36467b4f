"""Launcher: run an UNMODIFIED script of the reference checkout (``full_eval.py``, ``full_run.py`` ...) with the
drop-in's ``source`` package in front of the reference's.

    cd /path/to/points2surf
    python -m points2surf_amd.dropin.run full_eval.py --indir datasets --outdir results ...
    torchrun --nproc-per-node 8 -m points2surf_amd.dropin.run full_eval.py ...

**All visible GPUs by default -- for ``full_eval.py``.**  The reference wraps its model in ``torch.nn.DataParallel``
without ``device_ids`` (source/points_to_surf_eval.py:168): ``python full_eval.py`` on an 8-GPU node uses all 8.  The
launcher keeps that: started WITHOUT a torchrun environment (no ``WORLD_SIZE``) on a node with more than one visible
device it re-executes itself under ``torch.distributed.run`` with one rank per device (127.0.0.1 rendezvous, a free
port); the ranks shard the shapes (points2surf_amd/sharding.py).  ``P2S_GPUS=<n>`` picks the number of ranks
(``P2S_GPUS=1``: stay in this process; more than the visible devices is an error), ``HIP_VISIBLE_DEVICES`` the devices.
Only scripts whose every stage is rank-aware are ever started as several ranks (``RANK_AWARE_SCRIPTS``: ``full_eval.py``
-- the drop-in's eval, mesh and comparison stages shard or run on rank 0).  Anything else -- ``full_run.py`` calls the
reference's own ``points_to_surf_train``, ``make_dataset.py`` writes a data set -- runs in ONE process: N copies of a
training would all write ``models/<name>_model.pth`` and the logs.  ``P2S_GPUS=<n>`` with such a script is refused, and so
is such a script under torchrun with more than one rank (``WORLD_SIZE`` > 1).

Why a launcher: ``python full_eval.py`` puts the script's directory at ``sys.path[0]``, i.e. BEFORE anything on
``PYTHONPATH``, so ``from source import points_to_surf_eval`` would find the reference's own ``source`` package first.
Here ``sys.path`` becomes [drop-in, script directory, ...] and the script runs under ``runpy`` as ``__main__`` --
the file itself is not touched.  The reference modules need ``np.int`` (removed in numpy 1.24): restored as an alias.
"""
import os
import runpy
import sys


RANK_AWARE_SCRIPTS = ('full_eval.py',)      # every stage shards over the ranks or runs on rank 0 only
# the reference's other top-level scripts: training, data-set generation, conversion -- every rank would do ALL the work and
# write the same files.  Refused under torchrun with more than one rank (a user's own script that calls the drop-in's
# rank-aware functions is the user's business and runs)
SINGLE_PROCESS_SCRIPTS = ('full_run.py', 'full_train.py', 'make_dataset.py', 'make_pc_dataset.py', 'eval_dataset.py',
                          'dataset_for_deepsdf.py', 'blensor_script_template.py')


def ranks_to_spawn(environ=None, device_count=None, script=None):
    """how many ranks the launcher starts by itself: 0 = run in this process.  Under torchrun (WORLD_SIZE set) never;
    otherwise, for a rank-aware script, one per visible device, or P2S_GPUS of them.  ``script`` None: treated as
    rank-aware (callers that only ask about the environment)."""
    environ = os.environ if environ is None else environ
    if 'WORLD_SIZE' in environ:
        return 0
    want = environ.get('P2S_GPUS')
    if want is not None and int(want) <= 1:
        return 0
    if script is not None and os.path.basename(script) not in RANK_AWARE_SCRIPTS:
        if want is not None:
            raise SystemExit('points2surf_amd.dropin.run: P2S_GPUS=%s, but %s is not rank-aware (only %s are): its stages '
                             'would run once per rank and overwrite each other\'s files'
                             % (want, os.path.basename(script), ', '.join(RANK_AWARE_SCRIPTS)))
        return 0
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count() if torch.cuda.is_available() else 0
    n = device_count if want is None else int(want)
    if n > device_count:
        raise SystemExit('points2surf_amd.dropin.run: P2S_GPUS=%d but %d device(s) visible' % (n, device_count))
    return n if n > 1 else 0


def spawn_command(n, argv, port):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
            '--master-addr', '127.0.0.1', '--master-port', str(port), '-m', 'points2surf_amd.dropin.run'] + list(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit('points2surf_amd.dropin.run: no such script: %s' % argv[0])
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.path.basename(script) in SINGLE_PROCESS_SCRIPTS:
        # torchrun --nproc-per-node N -m points2surf_amd.dropin.run full_run.py: N trainings writing the same model files
        raise SystemExit('points2surf_amd.dropin.run: WORLD_SIZE=%s, but %s is not rank-aware (of the reference\'s scripts only '
                         '%s are): its stages would run once per rank and overwrite each other\'s files -- start it as ONE '
                         'process' % (os.environ['WORLD_SIZE'], os.path.basename(script), ', '.join(RANK_AWARE_SCRIPTS)))
    n = ranks_to_spawn(script=script)
    if n:
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        env['PYTHONPATH'] = repo_root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        print('points2surf_amd.dropin.run: %d visible devices -> %d ranks (P2S_GPUS=1 to stay in one process)' % (n, n),
              flush=True)
        return subprocess.call(spawn_command(n, [script] + argv[1:], port), env=env)
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(os.path.dirname(here))
    for p in (os.path.dirname(script), repo, here):
        while p in sys.path:
            sys.path.remove(p)
    sys.path[0:0] = [here, os.path.dirname(script), repo]
    for name in [m for m in sys.modules if m == 'source' or m.startswith('source.')]:
        del sys.modules[name]
    import numpy as np
    if not hasattr(np, 'int'):
        np.int = int                       # source/sdf.py:75 and friends (numpy < 1.24 spelling)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
