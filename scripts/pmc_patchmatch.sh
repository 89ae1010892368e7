#!/bin/bash
# Counter passes for the finest-level PatchMatch kernel inside one real 700x700 pair (counters only, one pass per set — the pool's gpurun
# refuses --pmc mixed with API traces). Output: gpurun_out/<tag>/pmc/p<i>/…counter_collection.csv; summarise with scripts/pmc_summary.py.
tag=${1:-pmc}
out=gpurun_out/$tag/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/p$i.log 2>&1
done
python scripts/pmc_summary.py $out "${2:-void k_pm_step<1, 1,}" | tee $out/summary.txt
find $out -name "*_kernel_trace.csv" -delete; find $out -name "c_counter_collection.csv" -delete
