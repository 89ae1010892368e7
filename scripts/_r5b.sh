#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_correspondence.py -x -q -k "vote" > gpurun_out/r5b/vote.log 2>&1; echo "vote rc=$?"; tail -12 gpurun_out/r5b/vote.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "levels_match or end_to_end or minimum_and or dev_seams" > gpurun_out/r5b/pipe.log 2>&1; echo "pipe rc=$?"; tail -12 gpurun_out/r5b/pipe.log
python scripts/vote_levels.py 2>&1 | tail -3
