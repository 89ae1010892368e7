#!/bin/bash
# The round's measurement pass, run on the GPU box through gpurun (scripts/collect_profiles.py turns its output into profiles/):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash scripts/final_measure.sh rN'
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-300 $out/bench.json
cp profiles/round2_pmc_patchmatch.json $out/pmc_fallback_before.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency-flag > $out/prof_bench.json 2> $out/prof.err
find $out/prof -name "*kernel_trace*" -delete; find $out/prof -name "*.db" -delete
timeout 300 python scripts/pm_modes.py 700 > $out/pm_modes.log 2>&1
timeout 300 python bench.py --workload pair1000 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_1000.json 2>/dev/null
timeout 300 python bench.py --workload pair256l5 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $out/bench_256l5.json 2>/dev/null
timeout 300 python bench.py --workload batch64 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_batch.json 2>/dev/null
timeout 300 python bench.py --workload mixed256 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_mixed.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --dist-backend gloo --device-override 0 --inflight 2 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_2rank_gloo.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
ls -la $out
