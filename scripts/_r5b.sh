#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_color.py tests/test_gpu_pipeline.py -x -q -k "local_color_transfer_stages or levels_match or degenerate or downscaled or hub_pass or in0_tar0" > gpurun_out/r5b/hub2.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r5b/hub2.log
python scripts/flat_probe.py 2>&1 | tail -4
