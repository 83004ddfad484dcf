cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ntt1
rm -rf $OUT; mkdir -p $OUT
export NTT_ONLY=${NTT_ONLY:-20x256}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc_valu -o ntt -- python tools/ntt_bench.py > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o ntt -- python tools/ntt_bench.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_wait -o ntt -- python tools/ntt_bench.py > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM -d $OUT/pmc_wc -o ntt -- python tools/ntt_bench.py > $OUT/pmc5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ntt -- python tools/ntt_bench.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o ntt -- python tools/ntt_bench.py > $OUT/pmc3.log 2>&1
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) > $OUT/rocprofv3_summary.txt 2>&1
grep -E "k_ntt_tile" $OUT/rocprofv3_summary.txt | cut -c1-30,80-200
