#!/bin/bash
# Riccati / LQ sweep: MFMA kernel vs the VALU kernel (TREPAMD_LQ_LEGACY=1): numpy parity tests + timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_discopt_device.py tests/test_discopt.py tests/test_gpu_discopt_puppet.py -m gpu -x -q 2>&1 | tail -4
python tools/time_lq.py
TREPAMD_LQ_LEGACY=1 python tools/time_lq.py
