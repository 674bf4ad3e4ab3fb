"""The C ABI is usable from plain C: include/f8net.h compiles as strict C99 and a C host (examples/host_demo.c) links against
libf8net.so, builds and plans a net without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'f8net_amd', 'libf8net.so')


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')
def test_c99_host_links_and_plans(tmp_path):
    assert os.path.exists(LIB), 'build libf8net.so first (__graft_entry__.build())'
    exe = str(tmp_path / 'host_demo')
    cmd = ['gcc', '-std=c99', '-pedantic', '-Wall', '-Wextra', '-Werror', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'host_demo.c'), '-L' + os.path.dirname(LIB), '-lf8net',
           '-Wl,-rpath,' + os.path.dirname(LIB), '-o', exe]
    subprocess.check_call(cmd)
    out = subprocess.check_output([exe], text=True, env=dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', '')))
    assert 'libf8net version' in out and 'launches' in out
    assert '_res' in out                  # the residual join rides in the second conv's epilogue
    assert 'add:' not in out
