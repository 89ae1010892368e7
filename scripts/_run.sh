set -x
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_cli.py tests/test_gpu_pipeline.py tests/test_gpu_color.py -x -q -m gpu > gpurun_out/r4a/pytest.txt 2>&1; tail -5 gpurun_out/r4a/pytest.txt
ONLY_FP32=1 timeout 600 python scripts/pm_modes.py 700 > gpurun_out/r4a/pm_modes.log 2>&1; cat gpurun_out/r4a/pm_modes.log
