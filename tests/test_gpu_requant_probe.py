"""The two requantisations of f8_device.h against int_op_only_fix_quant (reference: models/fix_quant_ops.py:99-112, clamp [0, 255]) for EVERY int32
value, on the device:
  mode 0  requant_u8x4 (v_cvt_f32_i32, v_mul_f32 by 2^-n, v_cvt_pk_u8_f32) against the NON-wrapping quotient (what the planner bounds: conv
          accumulators and the chain launches' int32 stream), every shift it is used for (1 .. 16);
  mode 1  requant_u8x4_int (v_bfe_u32, v_add3_u32, v_ashr_pk_u8_i32: integer only) against the reference's WRAPPING int32 form, both operand positions,
          shifts 1 .. 30.
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
    assert sorted(rows) == [(0, n) for n in range(1, 21)] + [(1, n) for n in range(1, 31)], r.stdout
    assert all(rows[0, n] == 0 for n in range(1, 17)), rows
    assert all(rows[1, n] == 0 for n in range(1, 31)), rows            # the integer form: exact for every value and shift
    # beyond 16 the float form is NOT exact (values from 2^24 on): the hosts must not select it there (kRequantU8MaxShift)
    assert all(rows[0, n] > 0 for n in range(17, 21)), rows
