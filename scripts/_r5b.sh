#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "hub_pass or in_flight" > gpurun_out/r5b/hub.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r5b/hub.log
python scripts/s1_levels.py 9 | tail -1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline --no-latency-flag 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('value %.2f single %.2f' % (d['value'], d['single_pair_ms']))"
