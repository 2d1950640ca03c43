#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06i; mkdir -p $o
for rep in 1 2; do
for v in default vprol; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  NTPOLY_AMD_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wrp-check > $o/bench_$v.json 2> $o/bench.err; echo "bench $v rc=$?"
  python3 -c "import json;d=json.load(open('$o/bench_$v.json'));print('$v',d['value'],d['roofline']['ms_per_launch'],d['config']['blocks_ms'])"
done
done
rm -rf $o/tp
NTPOLY_AMD_FORCE_RCCL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/tp -o run -- python3 bench.py --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline --no-wrp-check --n 32768 > $o/panel_bench.json 2> $o/panel.err
cp $o/tp/*kernel_trace.csv $o/panel_kernels.csv 2>/dev/null; rm -rf $o/tp
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06i/panel_kernels.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 2 steps: find tile kernel launches
idx=[i for i,r in enumerate(rows) if 'k_spgemm_tile' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b+1]:
    print("%8.1f us  +%6.1f us  %s" % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Kernel_Name'][:90]))
PY
python3 -c "import json;d=json.load(open('gpurun_out/r06i/panel_bench.json'));print('panel step ms',d['ms_per_step'])"
