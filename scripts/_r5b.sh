#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_color.py -x -q -k "knn" > gpurun_out/r5b/knn.log 2>&1; echo "knn rc=$?"; tail -5 gpurun_out/r5b/knn.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "levels_match or downscaled" > gpurun_out/r5b/pipe.log 2>&1; echo "pipe rc=$?"; tail -3 gpurun_out/r5b/pipe.log
bash scripts/kernel_times.sh r5g "k_knn" in4_tar4_2 2>&1 | grep -v "run_\|entr\|cell" | tail -6
bash scripts/kernel_times.sh r5h "k_knn" 700 2>&1 | grep -v "run_\|entr\|cell" | tail -6
