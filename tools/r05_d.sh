#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05d; mkdir -p $o
for w in 12 16; do for od in 0 1; do echo "== waves $w order $od"; NTPOLY_AMD_T2_ORDER=$od NTPOLY_AMD_T2_WAVES=$w T2_ONLY=1 timeout 600 python3 tools/t2_ablate.py; done; done
NTPOLY_AMD_T2_ORDER=0 NTPOLY_AMD_T2_WAVES=16 NTPOLY_AMD_T2_STAMPS=$o/stamps.bin T2_ONLY=2 timeout 600 python3 tools/t2_ablate.py > /dev/null
python3 tools/t2_stamps.py $o/stamps.bin 60
NTPOLY_AMD_T2_ORDER=1 NTPOLY_AMD_T2_WAVES=12 NTPOLY_AMD_T2_STAMPS=$o/stamps.bin T2_ONLY=2 timeout 600 python3 tools/t2_ablate.py > /dev/null
python3 tools/t2_stamps.py $o/stamps.bin 60
