# prev.so vs cur.so inside one GPU call: steps at several sizes (tools/ab_lib.py) and isolated small launches warm / cold (tools/tiny_probe.py)
cd $GRAFT_REPO_ROOT
for p in ${SIZES:-256 1024 64}; do
  echo "== P=$p"; env P=$p timeout 600 python tools/ab_lib.py ${REPS:-3} build_ab/prev.so build_ab/cur.so 2>&1 | tail -2 | cut -c1-150
done
for cfg in ${SMALL:-1,128 1,512 8,128}; do
  IFS=, read p n <<< "$cfg"
  for l in prev cur prev cur; do echo -n "$l "; TMX_LIB=$PWD/build_ab/$l.so P=$p N=$n timeout 300 python tools/tiny_probe.py 2>&1 | tail -1 | cut -c1-200; done
done
