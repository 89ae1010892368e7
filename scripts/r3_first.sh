#!/bin/bash
# Round 3, first GPU pass: new parity tests, S2 rtol sweep, MFMA-busy counters of the current conv kernel, PMC tables of the coarser PatchMatch instantiations.
tag=${1:-r3a}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 900 python scripts/wls_rtol_sweep.py > $out/wls_rtol_sweep.json 2> $out/wls_rtol_sweep.err; tail -4 $out/wls_rtol_sweep.err
# VGG MFMA utilisation of the shipped conv kernel (two counter-only passes)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out/vgg/p1 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace -d $out/vgg/p2 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p2.log 2>&1
python scripts/pmc_by_grid.py $out/vgg "void k_conv3x3_mfma" > $out/vgg_mfma_by_grid.txt 2>&1; cat $out/vgg_mfma_by_grid.txt | head -40
# PatchMatch: all instantiations of one real pair, per kernel name
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc/p$i -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/pmc_p$i.log 2>&1
done
for pre in "void k_pm_step<1, 1," "void k_pm_step<2, 1," "void k_pm_step<4, 0," "void k_pm_step<8, 0,"; do echo "== $pre"; python scripts/pmc_summary.py $out/pmc "$pre"; done > $out/pmc_pm_all.txt 2>&1
# fabric bytes of every instantiation
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmcfw/p_$set -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/pmcfw_$set.log 2>&1
done
for pre in "void k_pm_step<1, 1," "void k_pm_step<2, 1," "void k_pm_step<4, 0," "void k_pm_step<8, 0,"; do echo "== $pre"; python scripts/pmc_summary.py $out/pmcfw "$pre"; done >> $out/pmc_pm_all.txt 2>&1
# keep per-kernel trace durations of the PMC run for the level kernels, drop the bulky CSVs
find $out -name "*_kernel_trace.csv" -delete; find $out -name "c_counter_collection.csv" -size +20M -delete
cat $out/pmc_pm_all.txt | head -120
ls -la $out
