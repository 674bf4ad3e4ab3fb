"""The float requantisations of f8_device.h against the integer form of int_op_only_fix_quant (reference: models/fix_quant_ops.py:99-112,
clamp [0, 255]) for EVERY int32 value and every shift the library uses them for (1 .. 16), on the device:
  mode 0  requant_u8x4 (v_cvt_f32_i32, v_mul_f32 by 2^-n, v_cvt_pk_u8_f32) against the NON-wrapping quotient (bounded conv accumulators);
  mode 1  requant_u8x4_wrap (v_add_u32 2^(n-1), v_cvt_f32_i32, v_fma_f32, v_cvt_pk_u8_f32) against the reference's WRAPPING int32 form.
The binary is built by f8net_amd/csrc/build.sh from tools/ubench/cvt_u8_probe.hip."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_float_requant_equals_integer_requant_for_every_int32():
    exe = os.path.join(ROOT, 'tools', 'ubench', 'cvt_u8_probe.bin')
    assert os.path.exists(exe), 'run f8net_amd/csrc/build.sh (it builds the probe)'
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    rows = {(int(m.group(1)), int(m.group(2))): int(m.group(3)) for m in re.finditer(r'mode=(\d) n=\s*(\d+) mismatches=(\d+)', r.stdout)}
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(rows) == [(m, n) for m in (0, 1) for n in range(1, 21)], r.stdout
    assert all(rows[m, n] == 0 for m in (0, 1) for n in range(1, 17)), rows
    # beyond 16 the float forms are NOT exact (values from 2^24 on): the hosts must not select them there (kRequantU8MaxShift)
    assert all(rows[m, n] > 0 for m in (0, 1) for n in range(17, 21)), rows
