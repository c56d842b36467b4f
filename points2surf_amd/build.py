"""Build libp2s_hip.so (gfx950) in-tree with hipcc.  ``python -m points2surf_amd.build``"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libp2s_hip.so')


def hipcc_path():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isfile(c) or c == 'hipcc'):
            return c
    return 'hipcc'


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
           '-o', LIB] + list(extra_flags) + os.environ.get('P2S_EXTRA_HIPCC_FLAGS', '').split() + sources()
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
