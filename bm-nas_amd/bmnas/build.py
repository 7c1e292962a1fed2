"""Build libbmnas_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
LIB = os.path.join(HERE, 'libbmnas_hip.so')
SOURCES = ['mixsum.hip', 'layernorm.hip', 'sdpa.hip', 'conv1x1.hip', 'bnmix.hip', 'linear.hip', 'adam.hip']


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm, /opt/rocm/bin/hipcc)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'bmnas_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source into bmnas/libbmnas_hip.so (in-tree, travels with gpurun)."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-unused-result', '-o', LIB + '.tmp'] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
