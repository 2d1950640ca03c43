#!/bin/bash
# the band search's kernel launched plainly instead of cooperatively (libntpoly_amd_plainbfs.so, an experiment build): the relabelled operand on two ranks sharing one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06av; mkdir -p $o
NTPOLY_AMD_LIB=$PWD/ntpoly_amd/libntpoly_amd_plainbfs.so NTPOLY_AMD_SHM_MB=1024 timeout 300 python3 tools/scope_diag.py 2 262144 > $o/diag2.log 2>&1; echo "rc=$?"
grep "^iterations" $o/diag2.log | cut -c1-60
