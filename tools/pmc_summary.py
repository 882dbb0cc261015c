#!/usr/bin/env python3
"""Merge rocprofv3 PMC passes (one results.db per pass directory) into a per-kernel table + derived numbers.

    pmc_summary.py <dir with p*/**/results.db> <batch> <L>  > profiles/xxx.txt     (also writes <dir>/traffic.json, alu.json)

bench.py imports collect() / derive() to turn the passes it runs itself (tools/pmc_workload under rocprofv3, inside the
benchmark run) into `roofline.traffic` and `roofline.alu`.

Counter units (MI355X_MICROARCH.md): SQ_* cycle counters in quad-cycles; FETCH_SIZE / WRITE_SIZE in KiB, taken at the L2's
memory-side (fabric) port -- so they are L2-MISS-side bytes and still include what the 256 MiB Infinity Cache serves; on
gfx950 FETCH_SIZE reports 1/2 of the bytes of these kernels' reads (calibrated for their access shapes with
tools/fetch_calib.hip), hence the x2; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import glob
import json
import sqlite3
import sys

N = 16384


def collect(root):
    """{kernel: {counter: average per dispatch}}, {kernel: [average duration in us, one per pass]}"""
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
    return vals, dur


def derive(vals, dur, batch, L, simds=1024):
    """keyswitch pipeline totals per keyswitch from the per-dispatch averages of its kernels (every dispatch = one chunk of
    `batch` keyswitches): L2-miss-side bytes, VALU wave-instructions, shader clock, per-kernel FP64-issue fractions"""
    ks = [k for k in vals if k.startswith("k_ks")]
    out = {"batch": batch, "L": L, "kernels": {}}
    tot_bytes = tot_valu = clk_num = clk_den = 0.0
    have_bytes = have_valu = False
    for k in ks:
        v, d = vals[k], sum(dur[k]) / len(dur[k])
        e = {"avg_us_under_pmc": d}
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["read_bytes"], e["write_bytes"] = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
            tot_bytes += e["read_bytes"] + e["write_bytes"]
            have_bytes = True
        if "SQ_INSTS_VALU" in v:
            e["valu_wave_instructions"] = v["SQ_INSTS_VALU"]
            tot_valu += v["SQ_INSTS_VALU"]
            have_valu = True
        if "GRBM_GUI_ACTIVE" in v:
            e["shader_clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / (d * 1e3)
            clk_num += v["GRBM_GUI_ACTIVE"] / 8
            clk_den += d * 1e3
        if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_ANY" in v:
            wc = v["SQ_WAVE_CYCLES"]
            e["wave_time_split"] = {"active": v["SQ_ACTIVE_INST_ANY"] / wc, "issue_stall": v["SQ_WAIT_INST_ANY"] / wc,
                                    "waitcnt_barrier": v["SQ_WAIT_ANY"] / wc}
        out["kernels"][k] = e
    out["traffic_bytes_per_keyswitch"] = tot_bytes / batch if have_bytes else None
    out["valu_wave_instructions_per_keyswitch"] = tot_valu / batch if have_valu else None
    # The clock: GRBM_GUI_ACTIVE / 8 XCDs / duration of the LONGEST kernel. For a short dispatch the counter also covers the set-up
    # around it (round 4: k_ksx_special, 165 us, read "3.02 GHz" on a 2.4 GHz part, which made its issue fraction read 0.44 instead of
    # 0.63); the ~1 ms k_ksx_main dispatch is long enough. Per-kernel clocks outside (0.5, 2.45) GHz are replaced by it.
    clocks = {k: e["shader_clock_ghz"] for k, e in out["kernels"].items() if "shader_clock_ghz" in e}
    longest = max(clocks, key=lambda k: out["kernels"][k]["avg_us_under_pmc"]) if clocks else None
    out["shader_clock_ghz"] = clocks[longest] if longest else (clk_num / clk_den if clk_den else None)
    for k, e in out["kernels"].items():
        if "valu_wave_instructions" in e and out["shader_clock_ghz"]:
            c = e.get("shader_clock_ghz")
            if c is None or not (0.5 < c < 2.45):
                e["shader_clock_ghz_raw"], c = c, out["shader_clock_ghz"]
                e["shader_clock_ghz"] = c
            # time the kernel's VALU instructions need at 4 cycles each if no SIMD ever idled / its measured duration
            e["fp64_issue_frac"] = e["valu_wave_instructions"] * 4 / simds / (c * 1e3) / e["avg_us_under_pmc"]
    out["alg_bytes_per_keyswitch"] = (L + 4 * L) * N * 8
    return out


NTT_ALG_BYTES = 2 * N * 8          # read + write of the polynomial; tables shared by the batch


def derive_ntt(vals, dur, batch=1024, simds=1024, clock_ghz=None):
    """The standalone _NTT / _INTT launches of the same passes (tools/pmc_workload ... 1: `batch` polynomials per launch, N = 16384, the
    exact FP64 fast path): per direction the transform kernel + the table-preparation kernel that runs in front of EVERY launch --
    L2-miss-side bytes and VALU wave-instructions per launch and per transform, duration under the counters, FP64-issue fraction.
    A 70 us dispatch is too short for its own GRBM_GUI_ACTIVE reading (pmc_summary.derive): `clock_ghz` = the keyswitch passes' clock."""
    out = {}
    prep = [k for k in vals if k.startswith("k_ntt_prepare")]
    for name, pref in (("fwd", "k_ntt_fwd_"), ("inv", "k_ntt_inv_")):
        ks = [k for k in vals if k.startswith(pref)]
        if not ks:
            continue
        k = max(ks, key=lambda kk: sum(dur.get(kk, [0.0])))
        v, d = vals[k], sum(dur[k]) / len(dur[k])
        e = {"kernel": k, "avg_us_under_pmc": d, "batch": batch, "alg_bytes_per_launch": batch * NTT_ALG_BYTES}
        if prep:
            pv, pd = vals[prep[0]], sum(dur[prep[0]]) / len(dur[prep[0]])
            e["prepare_kernel_avg_us_under_pmc"] = pd
        else:
            pv, pd = {}, 0.0
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["read_bytes_per_launch"] = (v["FETCH_SIZE"] + pv.get("FETCH_SIZE", 0.0)) * 1024 * 2
            e["write_bytes_per_launch"] = (v["WRITE_SIZE"] + pv.get("WRITE_SIZE", 0.0)) * 1024
            e["traffic_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
            e["traffic_over_algorithmic"] = e["traffic_bytes_per_launch"] / e["alg_bytes_per_launch"]
        if "SQ_INSTS_VALU" in v:
            e["valu_wave_instructions_per_launch"] = v["SQ_INSTS_VALU"] + pv.get("SQ_INSTS_VALU", 0.0)
            e["valu_wave_instructions_per_transform"] = v["SQ_INSTS_VALU"] / batch
            if clock_ghz:
                e["shader_clock_ghz"] = clock_ghz
                e["fp64_issue_frac_under_pmc"] = v["SQ_INSTS_VALU"] * 4 / simds / (clock_ghz * 1e3) / d
        if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_ANY" in v:
            wc = v["SQ_WAVE_CYCLES"]
            e["wave_time_split"] = {"active": v["SQ_ACTIVE_INST_ANY"] / wc, "issue_stall": v["SQ_WAIT_INST_ANY"] / wc,
                                    "waitcnt_barrier": v["SQ_WAIT_ANY"] / wc}
        out[name] = e
    return out


def main():
    root, batch, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    vals, dur = collect(root)
    print(f"# PMC summary (averages per dispatch); keyswitch batch {batch}, L={L}, N=16384; NTT batch 1024")
    print("# SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE/WRITE_SIZE in KiB, L2-miss side (they")
    print("# include Infinity Cache hits); the x2 is the gfx950 correction (FETCH_SIZE reports 1/2 of these kernels' reads)")
    names = sorted(vals, key=lambda k: -sum(dur.get(k, [0])))
    for k in names:
        v = vals[k]
        d = sum(dur[k]) / len(dur[k])
        print(f"\n{k}   avg duration under PMC {d:.1f} us")
        for c in sorted(v):
            print(f"    {c:24s} {v[c]:.6g}")
        clk = None
        if "GRBM_GUI_ACTIVE" in v:
            clk = v["GRBM_GUI_ACTIVE"] / 8 / (d * 1e3)
            print("    -> shader clock while it ran: %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)" % clk)
        if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_ANY" in v:
            wc = v["SQ_WAVE_CYCLES"]
            print("    -> wave time split: active %.1f%%  issue-stall %.1f%%  waitcnt/barrier %.1f%%" % (
                100 * v["SQ_ACTIVE_INST_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc, 100 * v["SQ_WAIT_ANY"] / wc))
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
            print("    -> VALU instructions per wave: %.0f" % (v["SQ_INSTS_VALU"] / v["SQ_WAVES"]))
        if "SQ_INSTS_VALU" in v and clk:
            if not (0.5 < clk < 2.45):
                print("    -> (that clock is not credible for a %.0f us dispatch -- the counter covers the set-up around it; see the summary below)" % d)
            else:
                print("    -> FP64-issue fraction: %.3f (VALU wave-instructions x 4 cycles / 1024 SIMDs / clock / duration)" % (
                    v["SQ_INSTS_VALU"] * 4 / 1024 / (clk * 1e3) / d))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
            hit = ""
            if "TCC_HIT_sum" in v:
                hit = "; L2 hit rate %.1f%%" % (100 * v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]))
            print("    -> L2-miss-side bytes per dispatch: read %.3f GB (x2-corrected), write %.3f GB%s" % (rd / 1e9, wr / 1e9, hit))
    d = derive(vals, dur, batch, L)
    for k, e in d["kernels"].items():
        if "fp64_issue_frac" in e:
            print("# %s: FP64-issue fraction %.3f at %.2f GHz%s" % (k, e["fp64_issue_frac"], e["shader_clock_ghz"],
                  " (clock of the longest kernel; its own GRBM reading was %.2f GHz)" % e["shader_clock_ghz_raw"] if "shader_clock_ghz_raw" in e else ""))
    alg = d["alg_bytes_per_keyswitch"]
    if d["traffic_bytes_per_keyswitch"]:
        print(f"\n# keyswitch pipeline: measured L2-miss-side traffic {d['traffic_bytes_per_keyswitch'] / 1e6:.2f} MB per keyswitch "
              f"vs algorithmic {alg / 1e6:.2f} MB  (ratio {d['traffic_bytes_per_keyswitch'] / alg:.2f})")
    clk = d["shader_clock_ghz"] or 2.0
    if d["valu_wave_instructions_per_keyswitch"]:
        print(f"# keyswitch pipeline: {d['valu_wave_instructions_per_keyswitch']:.0f} VALU wave-instructions per keyswitch; at 4 cycles "
              f"each on 1024 SIMDs and {clk:.2f} GHz that is {d['valu_wave_instructions_per_keyswitch'] * 4 / 1024 / (clk * 1e3):.2f} us "
              f"of FP64 issue per keyswitch")
    json.dump({"valu_wave_instructions_per_keyswitch": d["valu_wave_instructions_per_keyswitch"], "shader_clock_ghz": clk,
               "batch": batch, "L": L, "per_kernel": {k: e.get("fp64_issue_frac") for k, e in d["kernels"].items()},
               "note": "sum of SQ_INSTS_VALU over the keyswitch kernels of one chunk / batch; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration"},
              open(f"{root}/alu.json", "w"))
    json.dump({"keyswitch_traffic_bytes_per_unit": d["traffic_bytes_per_keyswitch"], "batch": batch, "L": L,
               "alg_bytes_per_unit": alg,
               "note": "sum over the keyswitch kernels of 2*FETCH_SIZE + WRITE_SIZE (KiB->B), per keyswitch; L2-miss side, includes Infinity Cache hits"},
              open(f"{root}/traffic.json", "w"))


if __name__ == "__main__":
    main()
