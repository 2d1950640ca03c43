#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06aq; mkdir -p $o
NTPOLY_AMD_SHM_MB=1024 timeout 300 python3 tools/scope_diag.py 2 262144 natural > $o/diag2n.log 2>&1; echo "rc=$?"; grep "^iterations" $o/diag2n.log | cut -c1-120
NTPOLY_AMD_SHM_MB=1024 timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -o run -- python3 tools/scope_diag.py 1 262144 > $o/diag1.log 2>&1; echo "rc=$?"; grep "^iterations" $o/diag1.log | cut -c1-120
