#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -k "degenerate or hub_pass" > gpurun_out/r5b/deg.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r5b/deg.log
