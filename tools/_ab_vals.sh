# one knob, explicit values, several batch sizes: KNOB=TMX_PHASE1_MAX VALS="0 1000000" SIZES="24 32 64" bash tools/_ab_vals.sh
cd $GRAFT_REPO_ROOT
for p in ${SIZES:-32 64}; do
  for v in $VALS; do
    echo -n "P=$p $KNOB=$v  "; env P=$p $KNOB=$v timeout 300 python tools/ab_lib.py ${REPS:-2} tendermintx_amd/libtmx.so 2>&1 | tail -1 | cut -c30-200
  done
done
