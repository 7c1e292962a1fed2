"""Build libbmnas_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Each csrc/*.hip is compiled to its own object (in parallel, cached under csrc/.obj by source and
header mtimes) and the objects are linked into bmnas/libbmnas_hip.so, in-tree, so that the
library travels with a gpurun snapshot."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
OBJ = os.path.join(CSRC, '.obj')
LIB = os.path.join(HERE, 'libbmnas_hip.so')
SOURCES = ['mixsum.hip', 'layernorm.hip', 'sdpa.hip', 'conv1x1.hip', 'bnmix.hip', 'linear.hip', 'adam.hip',
           'head.hip', 'comm.hip', 'probe.hip', 'dropout.hip', 'pool.hip', 'mixconv.hip', 'lazyln.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-pass-failed']
# timing builds only (e.g. BMNAS_HIPCC_EXTRA=-DBMNAS_BODY_PROBES=1 python -m bmnas.build --force)
FLAGS += os.environ.get('BMNAS_HIPCC_EXTRA', '').split()


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm, /opt/rocm/bin/hipcc)')


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h'))]
    hs.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'bmnas_hip.h'))
    return [h for h in hs if os.path.exists(h)]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


STAMP = os.path.join(OBJ, '.flags')


def _flags_changed():
    """The compile flags of the objects in .obj (a probe build — BMNAS_HIPCC_EXTRA — must not survive into later
    normal runs, nor the other way round)."""
    try:
        with open(STAMP) as f:
            return f.read() != ' '.join(FLAGS)
    except OSError:
        return True


def needs_build():
    if not os.path.exists(LIB) or _flags_changed():
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(hipcc, src, force, verbose):
    obj = os.path.join(OBJ, src + '.o')
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in [path] + _headers()):
            return obj
    cmd = [hipcc] + FLAGS + ['-c', path, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(force=False, verbose=False, jobs=None):
    """Compile every HIP source and link bmnas/libbmnas_hip.so.  force=True recompiles every
    object (the driver's "does it build" check); otherwise objects newer than their sources and
    the headers are reused."""
    if not force and not needs_build():
        return LIB
    hipcc = hipcc_path()
    os.makedirs(OBJ, exist_ok=True)
    if _flags_changed():
        force = True                       # objects compiled with other flags are stale whatever their mtimes say
    srcs = _sources()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(hipcc, s, force, verbose), srcs))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + objs
    cmd += ['-ldl']          # csrc/comm.hip binds RCCL lazily (dlopen): no load-time dependency on librccl
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    with open(STAMP, 'w') as f:
        f.write(' '.join(FLAGS))
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))
