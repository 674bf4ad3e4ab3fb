#!/bin/bash
# Tuning helper: device assembly of f8_chain.hip + the spill map of its FAST instances.
cd /root/repo/f8net_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only $* -o /tmp/chain.s f8_chain.hip 2>&1 | grep -E "error" -A3 | head
python3 /root/repo/tools/asm_spills.py /tmp/chain.s Lb1E | grep -v barriers
