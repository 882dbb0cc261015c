cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcg/p$i -- $R/tools/geom_bench > /dev/null 2>&1
done
python3 - <<'PY'
import glob, sqlite3, os
root=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/pmcg"
vals={}
for p in sorted(glob.glob(root+"/p*/**/*.db", recursive=True)):
    cur=sqlite3.connect(p).cursor()
    tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cc=[t for t in tabs if t.startswith("counters_collection")][0]
    for name,cname,n,avg in cur.execute(f"select kernel_name, counter_name, count(*), avg(value) from {cc} group by kernel_name, counter_name"):
        vals.setdefault(name[:60],{})[cname]=avg
for k,v in vals.items():
    print(k)
    for c in sorted(v): print("   %-26s %.6g"%(c,v[c]))
PY
