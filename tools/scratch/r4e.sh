mkdir -p gpurun_out/r4e; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4e/kt -- python $R/tools/batch_sweep.py 6 1 > $R/gpurun_out/r4e/trace.log 2>&1
python3 $R/tools/rocprof_summary.py $(find $R/gpurun_out/r4e/kt -name "*.db" | head -1) > $R/gpurun_out/r4e/lat_trace.txt 2>&1
python3 - <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/r4e/lat_trace.txt
import sqlite3, glob, os
db = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4e/kt/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
ks = [r for r in rows if "k_ksq" in r[0]]
# the last full keyswitch: four consecutive kernels
print("\n# timeline of the last lone keyswitch (us from the first kernel's start): name start end | gap to the previous kernel")
last = ks[-4:]
t0 = last[0][1]
prev = None
for n, s, e in last:
    print("%-28s %8.2f %8.2f | gap %.2f" % (n.split("<")[0], (s - t0) / 1e3, (e - t0) / 1e3, 0 if prev is None else (s - prev) / 1e3))
    prev = e
# distance between consecutive keyswitches
starts = [r[1] for r in ks if "k_ksq_intt<" in r[0]]
d = [(b - a) / 1e3 for a, b in zip(starts[-10:], starts[-9:])]
print("# keyswitch-to-keyswitch period (us), last launches:", ["%.1f" % x for x in d])
PY
rm -rf $R/gpurun_out/r4e/kt
cat $R/gpurun_out/r4e/lat_trace.txt | cut -c1-200 | head -40
cd $R
timeout 300 python tools/batch_sweep.py 6 1 2>&1 | grep batch
timeout 300 python tools/batch_sweep.py 7 1 2>&1 | grep batch
