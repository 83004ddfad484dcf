# one knob at several batch sizes, fresh process per measurement: KNOB=TMX_BASE_FIRST SIZES="16 32 64 128" bash tools/_ab_sizes.sh
cd $GRAFT_REPO_ROOT
for p in ${SIZES:-16 32 64 128}; do
  for v in 0 1; do
    echo -n "P=$p $KNOB=$v  "; env P=$p $KNOB=$v timeout 300 python tools/ab_lib.py ${REPS:-2} tendermintx_amd/libtmx.so 2>&1 | tail -1 | cut -c30-200
  done
done
