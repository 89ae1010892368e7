#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_correspondence.py tests/test_gpu_color.py tests/test_gpu_pipeline.py -q -k "dead or nan_matching" > gpurun_out/r5b/dead.log 2>&1; echo "dead rc=$?"; tail -30 gpurun_out/r5b/dead.log
