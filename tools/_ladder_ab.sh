# Level-2 ladder passes: parity first, then rows-per-inversion x segments inside one GPU call (fresh process per setting: the knobs are read once)
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_trace.py tests/test_commit_pipeline.py -m gpu -x -q 2>&1 | tail -3
TMX_FUZZ_TRACE_SEEDS=${TRACE_SEEDS:-12} timeout 900 python -m pytest tests/test_fuzz_extended.py -m gpu -x -q -k fuzz_trace_rows 2>&1 | tail -2
CFGS=${CFGS:-16,4 8,4 32,4 64,4 16,8 32,8 16,2}
for cfg in $CFGS; do
  IFS=, read r g sd <<< "$cfg"; sd=${sd:-1}
  echo "ROWS=$r SEGS=$g SIDE=$sd: $(TMX_TRACE_ROWS=$r TMX_TRACE_SEGS=$g TMX_TRACE_SIDE=$sd SECTIONS=ladders timeout 300 python tools/trace_bench.py 2>&1 | grep -i 'ladders' | tr -s ' ')"
done
