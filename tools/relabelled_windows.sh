#!/bin/bash
# The relabelled headline (bench.py --permute 42) in the two iteration windows the rounds have quoted it in -- iterations 4..13
# (--steps 10 --warmup 3) and 11..50 (--steps 40 --warmup 10) -- with the sources of rounds 3, 4 and 5 on the same box
# (VERDICT r4 weak 7).  The older trees are exported and built under _cmp/<commit>/ (git archive; python -m ntpoly_amd._build).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tree in _cmp/3cf4f3c _cmp/30ec524 .; do
  for win in "10 3" "40 10"; do
    set -- $win
    line=$(cd $tree && timeout 600 python3 bench.py --permute 42 --steps $1 --warmup $2 --no-cpu-baseline --no-wrp-check 2>/dev/null | tail -1)
    echo "$tree steps=$1 warmup=$2 $(echo "$line" | grep -o '"value": [0-9.e+]*' | head -1) $(echo "$line" | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
