#!/bin/bash
# two ranks (shared-memory test transport) solving the relabelled headline operand; rank 1 under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06as; mkdir -p $o
name=s$$x
export WORLD_SIZE=2 LOCAL_RANK=0 NTPOLY_AMD_COMM=shm:$name NTPOLY_AMD_SHM_MB=1024
RANK=0 timeout 400 python3 tools/scope_diag.py 2 262144 > $o/rank0.log 2>&1 &
p0=$!
RANK=1 timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $o/prof -o run -- python3 tools/scope_diag.py 2 262144 > $o/rank1.log 2>&1
echo "rank1 rc=$?"; wait $p0; echo "rank0 rc=$?"
rm -f /dev/shm/ntpoly_amd_$name
grep "^iterations" $o/rank0.log | cut -c1-60
timeout 100 python3 tools/prof_summary.py $o/prof/run_results.db > $o/kernel_stats_rank1.csv; head -12 $o/kernel_stats_rank1.csv | cut -c1-130
python3 - <<'PY'
import sqlite3
c = sqlite3.connect("gpurun_out/r06as/prof/run_results.db")
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
# the last 60 kernels of the run (inside the last solve's iterations): names, durations, gaps
for n, s, e in rows[-60:]:
    import re
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_\w+|rocprim::\w+)", n)
    print("%10.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, m.group(1) if m else n[:40]))
PY
rm -rf $o/prof
