cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
P=${P:-1} WARM=5 STEPS=5 rocprofv3 --kernel-trace -d gpurun_out/tl1 -o tl -- python tools/profile_step.py > gpurun_out/tl1.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/tl1/**/*.db', recursive=True)[0])
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
ded = [i for i, r in enumerate(rows) if "k_ed_dedup" in r[0]]
i1, i2 = ded[-2], ded[-1]
t0 = rows[i1][1]
for r in rows[i1-3:i2]:
    print(f"{r[0][:30]:30s} start {((r[1]-t0)/1e3):9.1f} us  dur {((r[2]-r[1])/1e3):8.1f} us  stream {r[3]}")
PY
