#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_switches.py -x -q -k "full_resolution or TAIL" 2>&1 | tail -2
bash tools/chain_ab.sh default "$@" default "$@"
