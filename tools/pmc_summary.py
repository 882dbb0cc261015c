#!/usr/bin/env python3
"""Merge rocprofv3 PMC passes (one results.db per pass directory) into a per-kernel table + derived numbers.
usage: pmc_summary.py <dir with p*/**/results.db> <batch> <L>  > profiles/xxx.txt ; also writes <dir>/traffic.json"""
import glob
import json
import sqlite3
import sys

root, batch, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vals, dur = {}, {}
for p in sorted(glob.glob(f"{root}/p*/**/*.db", recursive=True)):
    cur = sqlite3.connect(p).cursor()
    for name, cname, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                           "group by kernel_name, counter_name"):
        k = name.split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            vals.setdefault(k, {})[cname] = avg
    for name, n, avg in cur.execute("select name, count(*), avg(end-start) from kernels group by name"):
        k = name.split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            dur.setdefault(k, []).append(avg / 1e3)
print(f"# PMC summary (averages per dispatch); keyswitch batch {batch}, L={L}, N=16384; NTT batch 1024")
print("# SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE/WRITE_SIZE in KiB;")
print("# fetch_x2 applies the gfx950 correction (FETCH_SIZE reports 1/2 of wide coalesced reads)")
names = sorted(vals, key=lambda k: -sum(dur.get(k, [0])))
ks_bytes = 0.0
ks_valu = 0.0
clk_num = clk_den = 0.0
for k in names:
    v = vals[k]
    d = sum(dur[k]) / len(dur[k])
    print(f"\n{k}   avg duration under PMC {d:.1f} us")
    for c in sorted(v):
        print(f"    {c:24s} {v[c]:.6g}")
    if k.startswith("k_ks") and "SQ_INSTS_VALU" in v:
        ks_valu += v["SQ_INSTS_VALU"]
    if "GRBM_GUI_ACTIVE" in v:
        # effective shader clock under this kernel = busy cycles / duration (MI355X_MICROARCH.md, DVFS)
        # (the counter is summed over the 8 XCDs)
        print("    -> shader clock while it ran: %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % (v["GRBM_GUI_ACTIVE"] / 8 / (d * 1e3)))
        if k.startswith("k_ks"):
            clk_num += v["GRBM_GUI_ACTIVE"] / 8; clk_den += d * 1e3
    if "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print("    -> wave time split: active %.1f%%  issue-stall %.1f%%  waitcnt/barrier %.1f%%" % (
            100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_WAIT_ANY"] / wc))
        print("    -> VALU instructions per wave: %.0f" % (v["SQ_INSTS_VALU"] / v["SQ_WAVES"]))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        print("    -> HBM-side bytes per dispatch: read %.3f GB (x2-corrected), write %.3f GB; L2 hit rate %.1f%%" % (
            rd / 1e9, wr / 1e9, 100 * v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
        if k.startswith("k_ks"):
            ks_bytes += rd + wr
alg = (L + 4 * L) * 16384 * 8
print(f"\n# keyswitch pipeline: measured HBM-side traffic {ks_bytes / batch / 1e6:.2f} MB per keyswitch "
      f"vs algorithmic {alg / 1e6:.2f} MB  (ratio {ks_bytes / batch / alg:.2f})")
clk = clk_num / clk_den if clk_den else 2.0
print(f"# keyswitch pipeline: {ks_valu / batch:.0f} VALU wave-instructions per keyswitch; at 4 cycles each on 1024 SIMDs and "
      f"{clk:.2f} GHz that is {ks_valu / batch * 4 / 1024 / (clk * 1e3):.2f} us of FP64 issue per keyswitch")
json.dump({"valu_wave_instructions_per_keyswitch": ks_valu / batch, "shader_clock_ghz": clk, "batch": batch, "L": L,
           "note": "sum of SQ_INSTS_VALU over the keyswitch kernels of one chunk / batch; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration"},
          open(f"{root}/alu.json", "w"))
json.dump({"keyswitch_traffic_bytes_per_unit": ks_bytes / batch, "batch": batch, "L": L, "alg_bytes_per_unit": alg,
           "note": "sum over the five keyswitch kernels of 2*FETCH_SIZE + WRITE_SIZE (KiB->B), per keyswitch"},
          open(f"{root}/traffic.json", "w"))
