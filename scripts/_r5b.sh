#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_cli.py -x -q -s -k "natural_demo_pair_matches or cli_on_the_reference" > gpurun_out/r5b/natfix.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/r5b/natfix.log | tail -12
