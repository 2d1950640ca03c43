set -x
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_kernel_vs_oracle or ps_gemm_golden or local_gemm or config4" 2>&1 | tail -15
