#!/bin/bash
# round 5, first GPU call: the self-launching multi-rank bench, configs[3], issue-side counters of the headline kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05a; mkdir -p $o
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench1.json 2> $o/bench1.err; echo "bench1 rc=$?"
timeout 600 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $o/bench2_shm.json 2> $o/bench2_shm.err; echo "bench2 rc=$?"
timeout 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 > $o/bench4_shm.json 2> $o/bench4_shm.err; echo "bench4 rc=$?"
timeout 900 python3 bench.py --config 3 --steps 5 --warmup 2 --blocks 3 > $o/cfg3.json 2> $o/cfg3.err; echo "cfg3 rc=$?"
timeout 900 python3 bench.py --config 3 --permute 42 --steps 5 --warmup 2 --blocks 3 > $o/cfg3_perm.json 2> $o/cfg3_perm.err; echo "cfg3p rc=$?"
timeout 900 python3 bench.py --config 3 --gpus 2 --steps 5 --warmup 2 --blocks 3 > $o/cfg3_2.json 2> $o/cfg3_2.err; echo "cfg3x2 rc=$?"
NTPOLY_AMD_SHM_MB=2048 timeout 900 python3 bench.py --config 3 --gpus 2 --permute 42 --steps 3 --warmup 1 --blocks 1 > $o/cfg3_2_perm.json 2> $o/cfg3_2_perm.err; echo "cfg3x2p rc=$?"
tools/pmc_bench_passes.sh r05a/pmc_tile "" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"
python3 tools/pmc_summary.py gpurun_out/r05a/pmc_tile k_spgemm_tile > $o/pmc_tile.txt
rm -rf gpurun_out/r05a/pmc_tile
cat $o/pmc_tile.txt
for f in bench1 bench2_shm bench4_shm cfg3 cfg3_perm cfg3_2 cfg3_2_perm; do echo "== $f"; python3 - <<PY
import json
try:
    l=json.loads(open("$o/$f.json").read().strip().splitlines()[-1])
    print(l["value"], l["unit"], l["ms_per_step"], l["roofline"]["kernel"][:40], l["roofline"]["ms_per_launch"], l["config"].get("blocks_ms"))
except Exception as e:
    print("ERR", e); print(open("$o/$f.err").read()[-1500:])
PY
done
