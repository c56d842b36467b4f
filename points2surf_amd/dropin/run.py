"""Launcher: run an UNMODIFIED script of the reference checkout (``full_eval.py``, ``full_run.py`` ...) with the
drop-in's ``source`` package in front of the reference's.

    cd /path/to/points2surf
    python -m points2surf_amd.dropin.run full_eval.py --indir datasets --outdir results ...
    torchrun --nproc-per-node 8 -m points2surf_amd.dropin.run full_eval.py ...

Why a launcher: ``python full_eval.py`` puts the script's directory at ``sys.path[0]``, i.e. BEFORE anything on
``PYTHONPATH``, so ``from source import points_to_surf_eval`` would find the reference's own ``source`` package first.
Here ``sys.path`` becomes [drop-in, script directory, ...] and the script runs under ``runpy`` as ``__main__`` --
the file itself is not touched.  The reference modules need ``np.int`` (removed in numpy 1.24): restored as an alias.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit('points2surf_amd.dropin.run: no such script: %s' % argv[0])
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
