#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r06n; mkdir -p $o
for v in default vhalf; do
  lib=ntpoly_amd/libntpoly_amd_$v.so; [ $v = default ] && lib=ntpoly_amd/libntpoly_amd.so
  export NTPOLY_AMD_LIB=$PWD/$lib
  timeout 600 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum -d $o/p_$v -o run --output-format csv -- \
    python3 bench.py --config 3 --n 262144 --halfband 157 --steps 4 --warmup 2 --blocks 1 > $o/pmc_$v.log 2>&1
  python3 - <<PY
import csv,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/r06n/p_$v/run_counter_collection.csv')):
    if 'k_spgemm_tile' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$v', {k: '%.4g' % (sum(v[1:])/max(1,len(v)-1)) for k,v in acc.items()})
PY
  rm -rf $o/p_$v
done
