"""No packed fp32 arithmetic in the device code (mhhip/build.py: on the MI355X boxes of this pool ``v_pk_fma_f32`` /
``v_pk_mul_f32`` / ``v_pk_add_f32`` can return wrong values while ``v_mfma_f32_32x32x16_f16`` instructions are in flight on
the same SIMD -- from the same wave or from ANOTHER kernel's wave sharing the CU).  The library is built with both of
clang's vectorisers off; this test disassembles what was actually linked."""
import glob
import os
import re
import shutil
import subprocess

import pytest

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


def test_every_source_is_compiled_without_the_vectorisers():
    from mhhip import build
    for src in build.sources():
        flags = build._flags(src, extra=[])
        assert '-fno-slp-vectorize' in flags and '-fno-vectorize' in flags, src


def test_the_guard_fails_loudly_without_a_disassembler(monkeypatch):
    """ADVICE / VERDICT r04: the link-time gate used to return 0 packed instructions when llvm-objdump was missing"""
    from mhhip import build
    monkeypatch.setattr(os.path, 'exists', lambda p: False if 'llvm-objdump' in str(p) else os.path.lexists(p))
    with pytest.raises(RuntimeError, match='llvm-objdump not found'):
        build.packed_instructions(build.LIB)


def test_linked_device_code_has_no_packed_fp32_instruction(tmp_path):
    from mhhip import build
    assert os.path.exists(OBJDUMP), 'llvm-objdump of the ROCm toolchain not found: the packed-fp32 guard cannot be held'
    assert build.packed_instructions(build.build()) == 0
    so = shutil.copy(build.build(), tmp_path / 'lib.so')
    subprocess.run([OBJDUMP, '--offloading', str(so)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp_path)
    objs = sorted(glob.glob(str(tmp_path / 'lib.so.*gfx950*')))
    assert len(objs) >= 6, objs                           # one code object per source file with device code
    packed = re.compile(build.PACKED_FP32)      # the fp32 arithmetic forms the fault was seen with (mhhip/build.py)
    n_mfma = 0
    seen = ''
    for o in objs:
        asm = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', o], check=True, capture_output=True, text=True).stdout
        assert len(asm) > 1000, o
        hits = packed.findall(asm)
        assert not hits, (o, sorted(set(hits)), len(hits))
        n_mfma += asm.count('v_mfma_')
        seen += ' '.join(re.findall(r'<(\w*k_\w+)>:', asm))
    assert 'k_raster_strip' in seen and 'k_skin_fwd16' in seen and 'k_raster_grads' in seen
    assert n_mfma > 100        # the disassembly is the real one: the LBS kernels' matrix instructions are in it
