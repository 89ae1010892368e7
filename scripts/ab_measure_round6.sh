#!/bin/bash
# round 6, second pass: (a) level-0 hub pass blind vs host wait, (b) PatchMatch one persistent launch per level vs one per step (times + kernel trace + fabric counters),
# (c) fabric counters of the finest PatchMatch level on a demo photograph
tag=${1:-r6b}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for rep in 1 2; do
  for w in 0 1; do
    echo "== NCT_S1_HUB_WAIT=$w rep $rep"
    NCT_S1_HUB_WAIT=$w timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-natural --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: d.get(k) for k in ('value','single_pair_ms','single_pair_ms_min','single_pair_latency_flag_ms')})"
  done
done > $out/hub_wait_ab.txt 2>&1
cat $out/hub_wait_ab.txt
timeout 900 python scripts/pm_persist_ab.py 700 3 > $out/pm_persist_700.txt 2>&1; cat $out/pm_persist_700.txt
timeout 900 python scripts/pm_persist_ab.py 1000 2 > $out/pm_persist_1000.txt 2>&1; cat $out/pm_persist_1000.txt
for p in 0 1; do
  NCT_PM_PERSIST=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_persist$p -o k -- python scripts/pair_only.py 700 3 > $out/prof_persist$p.log 2>&1
  find $out/prof_persist$p -name "*kernel_trace*" -delete; find $out/prof_persist$p -name "*.db" -delete
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    NCT_PM_PERSIST=$p timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc_persist$p/p$i -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/pmc_persist${p}_p$i.log 2>&1
  done
done
for pre in "void k_pm_step<1, 1,|void k_pm_prop<1, 1," "void k_pm_level<1, 1," "void k_pm_level<2, 1," "void k_pm_level<4, 0," "void k_pm_level<8, 0,"; do
  for p in 0 1; do echo "== persist=$p  $pre"; python scripts/pmc_summary.py $out/pmc_persist$p "$pre"; done
done > $out/pmc_persist_summary.txt 2>&1
# demo photograph: finest PatchMatch level's fabric traffic (coherent NNF)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc_nat/p$i -o c --output-format csv -- python scripts/pair_only.py in1_tar1_2 1 > $out/pmc_nat_p$i.log 2>&1
done
python scripts/pmc_summary.py $out/pmc_nat "void k_pm_step<1, 1,|void k_pm_prop<1, 1," > $out/pmc_nat_summary.txt 2>&1
python scripts/pmc_per_dispatch.py $out/pmc_nat "void k_pm_step<1, 1,|void k_pm_prop<1, 1," > $out/pmc_nat_per_dispatch.txt 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "c_counter_collection.csv" -size +20M -delete
ls -la $out
