#!/bin/bash
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 GLOME_PARITY_LOG=gpurun_out/parity_levels.txt timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final_pytest.log
grep -q "failed\|error" gpurun_out/final_pytest.log && exit 1
bash tools/final_measure.sh r02_h
