#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06at; mkdir -p $o
HSA_ENABLE_INTERRUPT=0 NTPOLY_AMD_SHM_MB=1024 timeout 300 python3 tools/scope_diag.py 2 262144 > $o/diag2i.log 2>&1; echo "polling rc=$?"
grep "^iterations" $o/diag2i.log | cut -c1-60
