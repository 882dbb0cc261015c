#!/usr/bin/env python3
"""fetch_reconcile.py -- round 5: why did the bench line's key-stream figure (0.69 MB per keyswitch) and profiles/r04_bytes.json's
(5.09 MB) differ? Both are differences of 2 x FETCH_SIZE between an un-aliased pass and a pass with every key row aliased onto row 0 --
but bench.py took the un-aliased pass from the SHIPPED library (tools/pmc_workload) and the aliased one from the PROFILING build
(tools/pmc_workload_prof), while tools/byte_budget.py took both from the profiling build. This runs every combination on ONE box, twice,
with the raw L2 -> fabric request counters beside FETCH_SIZE, and prints bytes per keyswitch. `shipped_kernels_keys_aliased` is the
leg bench.py uses since round 5: libhexl_mi355x_keyalias.so = the shipped kernel objects, one launcher function differs.

    python tools/fetch_reconcile.py [--out DIR] [--L 7]
"""
import argparse
import json
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from byte_budget import run_pass

GROUPS = {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE", "ea": "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum", "l2": "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum",
          "tcp": "TCP_TCC_READ_REQ_sum"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "reconcile"))
    ap.add_argument("--L", type=int, default=7)
    ap.add_argument("--repeats", type=int, default=2)
    a = ap.parse_args()
    out = Path(a.out).resolve()
    out.mkdir(parents=True, exist_ok=True)
    batch = 256
    legs = [("shipped", ROOT / "tools" / "pmc_workload", {}),
            ("shipped_kernels_keys_aliased", ROOT / "tools" / "pmc_workload_keyalias", {"HEXL_KSX_ALIAS": "1"}),
            ("prof_mask0", ROOT / "tools" / "pmc_workload_prof", {"HEXL_KSX_ALIAS": "0"}),
            ("prof_keys_aliased", ROOT / "tools" / "pmc_workload_prof", {"HEXL_KSX_ALIAS": "1"})]
    res = {"L": a.L, "chunk": batch, "units": "per keyswitch, summed over k_ksx_intt + k_ksx_special + k_ksx_main", "legs": {}}
    for name, exe, env in legs:
        for rep in range(a.repeats):
            e = {}
            for grp, counters in GROUPS.items():
                try:
                    vals, dur = run_pass(out / f"{name}_{rep}_{grp}", counters, [str(exe), str(batch), str(a.L), "2"],
                                         dict(os.environ, TMPDIR="/tmp", **env))
                except Exception as ex:
                    e.setdefault("errors", []).append(str(ex)[:200])
                    continue
                for k, v in vals.items():
                    if not k.startswith("k_ksx"):
                        continue
                    for c, x in v.items():
                        e[c] = e.get(c, 0.0) + x / batch
                        e.setdefault("per_kernel", {}).setdefault(k.split("<")[0], {})[c] = x / batch
            if "FETCH_SIZE" in e:
                e["fabric_read_MB_2xFETCH_SIZE"] = e["FETCH_SIZE"] * 1024 * 2 / 1e6
            if "WRITE_SIZE" in e:
                e["fabric_write_MB_WRITE_SIZE"] = e["WRITE_SIZE"] * 1024 / 1e6
            if "TCC_EA_RDREQ_sum" in e:
                r32 = e.get("TCC_EA_RDREQ_32B_sum", 0.0)
                e["ea_read_MB_if_64B"] = ((e["TCC_EA_RDREQ_sum"] - r32) * 64 + r32 * 32) / 1e6
                e["ea_read_MB_if_128B"] = ((e["TCC_EA_RDREQ_sum"] - r32) * 128 + r32 * 32) / 1e6
            if "TCC_MISS_sum" in e:
                e["l2_miss_MB_at_128B"] = e["TCC_MISS_sum"] * 128 / 1e6
                e["l2_hit_rate"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
            res["legs"][f"{name}#{rep}"] = e
            print(name, rep, json.dumps({k: round(v, 3) for k, v in e.items() if isinstance(v, float)}), flush=True)
    (out / "r05_fetch_reconcile.json").write_text(json.dumps(res, indent=1))
    print("wrote", out / "r05_fetch_reconcile.json")


if __name__ == "__main__":
    main()
