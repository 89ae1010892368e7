out=gpurun_out/$1; mkdir -p $out/vgg; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out/vgg/p1 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CU_CYCLES --kernel-trace -d $out/vgg/p2 -o c --output-format csv -- python scripts/vgg_only.py > $out/vgg_p2.log 2>&1
python scripts/pmc_by_grid.py $out/vgg "void k_conv3x3_mfma" > $out/vgg_mfma_by_grid.txt 2>&1
grep "grid=" $out/vgg_mfma_by_grid.txt
