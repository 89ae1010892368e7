#!/bin/bash
# round 6 first pass: gpu tests, bench line, kernel trace
tag=r6a
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency-flag > $out/prof_bench.json 2> $out/prof.err
find $out/prof -name "*kernel_trace*" -delete; find $out/prof -name "*.db" -delete
ls -la $out $out/prof/*
