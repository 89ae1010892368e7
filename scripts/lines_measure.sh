#!/bin/bash
# EXPERIMENTAL S2 block step (NCT_S2_LINES=1) against the default cycle: bench throughput / single-pair latency on the 700x700 pair, stage clock on the demo photographs.
# usage (GPU box): bash scripts/lines_measure.sh <tag>      -> gpurun_out/<tag>/
out=gpurun_out/${1:-lines}; mkdir -p $out
for on in 0 1; do
  NCT_S2_LINES=$on timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_lines$on.json 2> $out/bench_lines$on.err
  NCT_S2_LINES=$on timeout 600 python scripts/natural_report.py 5 in1_tar1_2 in4_tar4_2 in0_tar0_2 > $out/natural_lines$on.md 2> $out/natural_lines$on.err
done
