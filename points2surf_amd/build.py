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
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inl')) + \
        glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def _flags(extra_flags=()):
    return ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function'] + \
        list(extra_flags) + os.environ.get('P2S_EXTRA_HIPCC_FLAGS', '').split()


def build(force=False, verbose=True, extra_flags=(), jobs=None):
    """one object per translation unit under <repo>/build/obj/ (rebuilt when the source, a header or the flags changed), linked
    into libp2s_hip.so; ``force`` recompiles everything"""
    import concurrent.futures
    import hashlib
    if not force and not needs_build():
        return LIB
    flags = _flags(extra_flags)
    objdir = os.path.join(os.path.dirname(HERE), 'build', 'obj')          # (build/ is git- and gpurun-ignored)
    os.makedirs(objdir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inl')) + \
        glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h'))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    tag = hashlib.sha1(' '.join(flags).encode()).hexdigest()[:10]
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, '%s.%s.o' % (os.path.basename(src)[:-4], tag))
        objs.append(obj)
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc_path()] + flags + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = jobs or int(os.environ.get('P2S_BUILD_JOBS', min(8, os.cpu_count() or 1)))
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-shared', '-fPIC', '-fno-gpu-rdc', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
