"""Compile the HIP sources into the in-tree C-ABI shared library (gfx950 only).

Every ``csrc/*.hip`` is compiled to its own object (in parallel, only when stale) and the objects are linked into
``mhhip/libmhmocap_hip.so``.  Per-file flags:

* EVERY file is built with ``-fno-slp-vectorize -fno-vectorize``: no packed fp32 arithmetic (``v_pk_fma_f32`` /
  ``v_pk_mul_f32`` / ``v_pk_add_f32``) anywhere in the library.  On the MI355X boxes of this pool a packed fp32 instruction
  can return WRONG values while ``v_mfma_f32_32x32x16_f16`` instructions are in flight on the same SIMD -- from the same
  wave (round 2: hipcc packed the fp32 epilogue that consumes the accumulators in ``mh_lbs.hip``; wrong values on lanes
  48-63 of some waves of some launches, tools/stress_lbs.py) AND from another kernel's wave that shares the CU (round 4:
  k_raster_strip, whose face staging hipcc had packed, lost ~1 face per 1000 bodies and launch whenever the frame-sharded
  run's neighbour-frame LBS forward ran beside it; tools/c4_probe.py).  Every input of the affected
  expressions was verified deterministic and correct; the same source without packed instructions is bit-stable.
  Nothing is lost: packed fp32 issues at half the rate of the plain instructions on this chip
  (profiles/r04_ubench_valu_lds.txt).  tests/test_build_flags.py holds the disassembly to zero packed fp32 instructions.
"""
import concurrent.futures
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
OBJ = os.path.join(os.path.dirname(HERE), 'build')
LIB = os.path.join(HERE, 'libmhmocap_hip.so')

COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-munsafe-fp-atomics',
          '-fno-slp-vectorize', '-fno-vectorize']
PER_FILE = {}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', '*.h')) + [os.path.abspath(__file__)]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in deps)


def _flags(src, extra=None):
    extra = os.environ.get('MHHIP_CXXFLAGS', '').split() if extra is None else extra
    return COMMON + PER_FILE.get(os.path.basename(src), []) + list(extra)


def _stamp(src):
    return _obj(src) + '.flags'


def _flags_changed(src, extra=None):
    """an object is only reused when it was compiled with exactly the flags this build would use (experiments through
    MHHIP_CXXFLAGS must not leave objects behind that a later plain build links silently -- a file compiled
    without -fno-slp-vectorize -fno-vectorize computes wrong values beside matrix instructions)"""
    try:
        with open(_stamp(src)) as f:
            return f.read() != ' '.join(_flags(src, extra))
    except OSError:
        return True


def is_stale():
    return _stale(LIB, sources() + _headers()) or any(_flags_changed(s) for s in sources())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> mhhip/libmhmocap_hip.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    extra = os.environ.get('MHHIP_CXXFLAGS', '').split()
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for src in sources():
        if force or _stale(_obj(src), [src] + hdrs) or _flags_changed(src, extra):
            jobs.append((src, [hipcc] + _flags(src, extra) + ['-c', src, '-o', _obj(src)]))

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)

    def compile_one(job):
        src, cmd = job
        if os.path.exists(_stamp(src)):
            os.remove(_stamp(src))
        run(cmd)
        with open(_stamp(src), 'w') as f:
            f.write(' '.join(_flags(src, extra)))

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(s) for s in sources()] + ['-o', LIB + '.tmp'])
    if os.environ.get('MHHIP_ALLOW_PACKED') != '1':            # (experiments set it: tools/mkvariant.sh builds do not come through here)
        try:
            n = packed_instructions(LIB + '.tmp')              # raises when there is no llvm-objdump to look with: the guard
        except Exception:                                      # must not pass silently
            os.remove(LIB + '.tmp')
            raise
        if n:
            os.remove(LIB + '.tmp')
            raise RuntimeError('%d packed arithmetic instructions (v_pk_*, v_pk_mov_b32 aside) in the linked device code: the fp32 forms return '
                               'wrong values beside matrix instructions on this hardware (see the module docstring); flags: %s'
                               % (n, ' '.join(extra) or '-'))
    os.replace(LIB + '.tmp', LIB)
    return LIB


# every packed ARITHMETIC instruction (the fault was characterised for the fp32 forms only -- not shown absent for
# the f16 / bf16 / integer ones, and the library has none of them either); v_pk_mov_b32 moves bits, v_cvt_pk_* are not v_pk_*
PACKED_FP32 = r'\bv_pk_(?!mov_b32\b)[a-z0-9_]+\b'


def objdump_path():
    """llvm-objdump of the toolchain that compiled the library; raises when there is none (the packed-fp32 guard must not
    pass silently on a host without it: ADVICE r04)"""
    cands = [os.path.join(os.path.dirname(os.path.realpath(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'))), '..', 'lib', 'llvm', 'bin',
                          'llvm-objdump'), '/opt/rocm/lib/llvm/bin/llvm-objdump']
    for c in cands:
        if os.path.exists(c):
            return c
    raise RuntimeError('llvm-objdump not found (looked in %s): the linked device code cannot be checked for packed fp32 '
                       'instructions; set MHHIP_ALLOW_PACKED=1 to link without the check' % ', '.join(cands))


def packed_instructions(lib):
    """number of packed fp32 ARITHMETIC instructions -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, the forms the fault was
    seen with (moves, packed integer / 16-bit forms and the conversions v_cvt_pk_* are not counted: the backend may emit them
    whatever the vectoriser flags say) -- in the gfx950 code objects of a linked library"""
    import re
    import shutil
    import tempfile
    objdump = objdump_path()
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        so = shutil.copy(lib, os.path.join(tmp, 'lib.so'))
        subprocess.run([objdump, '--offloading', so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        objs = glob.glob(os.path.join(tmp, 'lib.so.*gfx950*'))
        if not objs:
            raise RuntimeError('no gfx950 code object found in %s' % lib)
        for o in objs:
            asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', o], check=True, capture_output=True, text=True).stdout
            n += len(re.findall(PACKED_FP32, asm))
    return n


if __name__ == '__main__':
    print(build(force=True, verbose=True))
