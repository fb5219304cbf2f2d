#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "hidden or h256 or h512" 2>&1 | tail -6
