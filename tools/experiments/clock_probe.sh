cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
HEXL_KS_ONE_LANE=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $R/gpurun_out/clk -- python $R/tools/profile_ks.py 256 7 > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3, os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/clk"
for p in glob.glob(root+"/**/*.db", recursive=True):
    cur=sqlite3.connect(p).cursor()
    tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cc=[t for t in tabs if t.startswith("counters_collection")][0]
    for name,cname,n,avg,dur in cur.execute(f"select kernel_name, counter_name, count(*), avg(value), avg(end-start) from {cc} group by kernel_name, counter_name"):
        if "k_" in name: print(name[:40], cname, n, "avg=%.4g"%avg, "dur_us=%.1f"%(dur/1e3), "-> %.3f GHz (if per-XCD sum /8: %.3f)"%(avg/dur, avg/dur/8))
PY
rocm-smi --showclocks 2>/dev/null | head -20
