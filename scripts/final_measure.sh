#!/bin/bash
# The round's measurement pass, run on the GPU box through gpurun (scripts/collect_profiles.py turns its output into profiles/):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash scripts/final_measure.sh rN'
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1700 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-300 $out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency-flag --no-natural > $out/prof_bench.json 2> $out/prof.err
find $out/prof -name "*kernel_trace*" -delete; find $out/prof -name "*.db" -delete
timeout 300 python scripts/pm_modes.py 700 > $out/pm_modes.log 2>&1
timeout 300 python bench.py --workload pair1000 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_1000.json 2>/dev/null
timeout 300 python bench.py --workload pair256l5 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $out/bench_256l5.json 2>/dev/null
timeout 300 python bench.py --workload batch64 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_batch.json 2>/dev/null
timeout 300 python bench.py --workload mixed256 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_mixed.json 2>/dev/null
timeout 300 python scripts/mixed_batch_cli.py 32 4 > $out/cli_mixed.txt 2>&1; tail -1 $out/cli_mixed.txt
timeout 600 python scripts/cli_8gpu_shape.py 64 700 > $out/cli_8gpu_shape.txt 2>&1; tail -1 $out/cli_8gpu_shape.txt | cut -c1-300
timeout 300 python bench.py --gpus 2 --dist-backend gloo --device-override 0 --inflight 2 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $out/bench_2rank_gloo.json 2>/dev/null
# (round 6: the S2 kernels are unchanged since round 5 — the tolerance sweeps of profiles/round5_wls_rtol_*.{json,md} stand; SWEEP=1 repeats them)
[ -n "$SWEEP" ] && SWEEP_RTOLS=3e-8,1e-7,5e-8,2e-8,1e-8,1e-10 timeout 900 python scripts/wls_rtol_sweep.py > $out/wls_rtol_sweep.json 2> $out/wls_rtol_sweep.err
# counters: PatchMatch instantiations of one real pair (SQ / TA / TCP / TCC, fabric bytes), conv MFMA utilisation
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc/p$i -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/pmc_p$i.log 2>&1
done
for pre in "void k_pm_step<1, 1," "void k_pm_prop<1, 1," "void k_pm_step<2, 1," "void k_pm_prop<2, 1," "void k_pm_step<4, 0," "void k_pm_step<8, 0,"; do echo "== $pre"; python scripts/pmc_summary.py $out/pmc "$pre"; done > $out/pmc_pm_all.txt 2>&1
# colour-solver kernels out of the same six passes (round 4)
for pre in "void (anonymous namespace)::k_mg_block<6, false>" "void (anonymous namespace)::k_mg_down<6, 32, 14, float, false, true>" "void (anonymous namespace)::k_mg_up<6, 32, 16, float, false>" "void (anonymous namespace)::k_mg_block<6, true>" \
           "void (anonymous namespace)::k_mg_down<6, 32, 16, float, true" "void (anonymous namespace)::k_mg_up<6, 32, 16, float, true" \
           "void (anonymous namespace)::k_cg_apply" "void (anonymous namespace)::k_cg_update" "void k_s1_apply<true>" "void k_s1_update<false>" "k_s1_scal(" "k_s1_hub("; do
    echo "== $pre"; python scripts/pmc_summary.py $out/pmc "$pre"
done > $out/pmc_color_all.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out/vgg/p1 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace -d $out/vgg/p2 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p2.log 2>&1
python scripts/pmc_by_grid.py $out/vgg "void k_conv3x3_mfma" > $out/vgg_mfma_by_grid.txt 2>&1
python scripts/pmc_per_dispatch.py $out/pmc "void k_pm_step<1, 1,|void k_pm_prop<1, 1," > $out/pmc_pm_finest_per_dispatch.txt 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "c_counter_collection.csv" -size +30M -delete
# round 5: the reference's own demo inputs (natural photographs) through the stage clock, the kernel trace and the counters
for c in in0_tar0_2 in1_tar1_2 in2_tar2_2 in3_tar3_2 in4_tar4_0 in4_tar4_1 in4_tar4_2 in4_tar4_4 in4_tar4_8; do timeout 300 python scripts/natural_report.py 5 $c > $out/natural_$c.md 2> $out/natural_$c.err; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/nat_prof -o n -- python scripts/pair_only.py in4_tar4_2 2 > $out/nat_prof.log 2>&1
find $out/nat_prof -name "*kernel_trace*" -delete; find $out/nat_prof -name "*.db" -delete
timeout 300 python scripts/wls_natural_probe.py > $out/wls_natural_probe.txt 2>&1
[ -n "$SWEEP" ] && timeout 600 python scripts/wls_rtol_natural.py > $out/wls_rtol_natural.txt 2>&1
timeout 600 python scripts/natural_crcs.py > $out/natural_crcs.json 2> $out/natural_crcs.err        # all nine demo pairs: per-level CRCs of the GPU path (python scripts/natural_crcs.py check <file> compares them with the oracle's fixtures)
timeout 300 python scripts/flat_probe.py > $out/flat_probe.txt 2>&1
timeout 600 python scripts/stress_determinism.py 8 > $out/stress_determinism.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
ls -la $out
