#!/bin/bash
# one GPU-box visit: parity tests + smoke (+ optional extra command), logs under gpurun_out/
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout -k 10 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
