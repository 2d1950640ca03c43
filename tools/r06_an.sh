#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06an; mkdir -p $o
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_cdyn.so timeout 150 python3 -m pytest tests/test_gpu_complex_tile.py -q -m gpu -x > $o/tests.log 2>&1; echo "cdyn tests rc=$?"; tail -15 $o/tests.log | cut -c1-200
