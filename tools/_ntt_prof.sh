cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ntt
rm -rf $OUT; mkdir -p $OUT
python tools/ntt_bench.py 2>/dev/null | grep "^{" > $OUT/ntt_bench.jsonl
rocprofv3 --kernel-trace --stats -d $OUT/trace -o ntt -- python tools/ntt_bench.py > $OUT/trace.log 2>&1
export NTT_ONLY=20x256
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc_valu -o ntt -- python tools/ntt_bench.py > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o ntt -- python tools/ntt_bench.py > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o ntt -- python tools/ntt_bench.py > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o ntt -- python tools/ntt_bench.py > $OUT/pmc3.log 2>&1
python tools/rocpd_summary.py $(find $OUT -name "*.db" | sort) > $OUT/rocprofv3_summary.txt 2>&1
grep -E "k_ntt|k_lde|kernel " $OUT/rocprofv3_summary.txt | cut -c1-170 | head -40
