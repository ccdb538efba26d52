"""Compile the HIP sources into the in-tree C-ABI shared library (gfx950 only)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
LIB = os.path.join(HERE, 'libmhmocap_hip.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', '*.h'))
    return any(os.path.getmtime(f) > t for f in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> mhhip/libmhmocap_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
           '-Wno-unused-result', '-munsafe-fp-atomics'] + sources() + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
