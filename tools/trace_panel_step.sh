#!/bin/bash
# HIP API + kernel trace of the headline through a 1-rank RCCL communicator at N / 8 columns (what a panel step calls):
# the CSVs are left in gpurun_out/trace_panel/ for tools/trace_panel_step.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tp gpurun_out/trace_panel; mkdir -p gpurun_out/trace_panel
NTPOLY_AMD_FORCE_RCCL=1 timeout 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d gpurun_out/tp -o run -- python3 bench.py --steps 6 --warmup 3 --blocks 1 --no-cpu-baseline --no-wrp-check --n 32768 > /dev/null 2>&1
cp gpurun_out/tp/*hip_api_trace.csv gpurun_out/trace_panel/hip_api.csv
cp gpurun_out/tp/*kernel_trace.csv gpurun_out/trace_panel/kernels.csv
rm -rf gpurun_out/tp
ls -la gpurun_out/trace_panel
