#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06at; mkdir -p $o
SCOPE_DIAG_LONG_KERNEL=1 NTPOLY_AMD_SHM_MB=1024 timeout 400 python3 tools/scope_diag.py 2 262144 gather > $o/diag2k.log 2>&1; echo "rc=$?"
grep "^natural\|^one long" $o/diag2k.log | cut -c1-150; tail -2 $o/diag2k.log | cut -c1-200
