#!/bin/bash
mkdir -p gpurun_out/r5f
for c in in1_tar1_2 in4_tar4_2 in0_tar0_2; do
  timeout 300 python scripts/natural_report.py 5 $c > gpurun_out/r5f/natural_$c.md 2> gpurun_out/r5f/natural_$c.err; echo "$c rc=$?"
  sed -n 5,22p gpurun_out/r5f/natural_$c.md
done
