# stall picture of the two ladder passes of the Level-2 writer: bash tools/_ladder_pmc.sh   (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ldpmc*
export SECTIONS=ladders
for c in ${PMCS:-"SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_IFETCH"}; do :; done
IFS=';'
for c in ${PMCS:-SQ_INSTS_VALU SQ_INSTS_SALU;SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY;SQ_WAIT_INST_ANY SQ_WAIT_ANY;SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT;SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM;SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU;SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT}; do
  name=$(echo $c | tr ' ' '_')
  IFS=' '
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/ldpmc_$name -o t -- python tools/trace_bench.py > gpurun_out/ldpmc_$name.log 2>&1
  echo "== $c"
  python tools/rocpd_summary.py $(find gpurun_out/ldpmc_$name -name "*.db") 2>/dev/null | grep -i "k_trace_ladder" | cut -c1-60,80-200
  IFS=';'
done
