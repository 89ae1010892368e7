#!/bin/bash
# counter passes over one 700x700 pair for the colour-solver kernels (WLS V-cycle legs, Krylov kernels, S1 operator): SQ cycle split, LDS, memory path, fabric bytes
# usage (GPU box): bash scripts/pmc_color.sh <tag>
tag=${1:-pmc_color}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o c --output-format csv -- python scripts/pair_only.py 700 1 > $out/p$i.log 2>&1
done
for pre in "void (anonymous namespace)::k_mg_down<6, 32, 16, double" "void (anonymous namespace)::k_mg_up<6, 32, 16, double" "void (anonymous namespace)::k_mg_down<6, 32, 16, float" "void (anonymous namespace)::k_mg_up<6, 32, 16, float" \
           "void (anonymous namespace)::k_cg_apply" "void (anonymous namespace)::k_cg_update" "void k_s1_apply<true>" "void k_s1_update<false>" "k_s1_scal(" "k_s1_hub("; do
    echo "== $pre"; python scripts/pmc_summary.py $out "$pre"
done > $out/pmc_color_all.txt 2>&1
find $out -name "*_kernel_trace.csv" -delete; find $out -name "c_counter_collection.csv" -size +30M -delete
cat $out/pmc_color_all.txt
