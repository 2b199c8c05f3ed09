#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 --pmc counter_collection CSVs (one per pass).
Usage: pmc_counters_summary.py <csv> [<csv> ...] > summary.json
Kernel names are cut at the argument list; launches of one template instance are averaged together."""
import collections
import csv
import json
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for fi, path in enumerate(sys.argv[1:]):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            if name.startswith("void "):
                name = name[5:]
            c = row["Counter_Name"]
            acc[name][c] += float(row["Counter_Value"])
            cnt[name][c].add((fi, row["Dispatch_Id"]))      # a counter may sit in several passes: every (pass, dispatch) is one sample
    out = {}
    for k in sorted(acc):
        # launches: dispatches of ONE pass (a counter that sits in several passes has one sample per pass and dispatch)
        out[k] = {"launches": max(max(sum(1 for f, _ in v if f == fi) for fi in {f for f, _ in v}) for v in cnt[k].values())}
        for c in sorted(acc[k]):
            out[k][c] = acc[k][c] / max(len(cnt[k][c]), 1)
        d = out[k]
        # SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE sums the 8 XCDs
        if "SQ_INSTS_VALU" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            d["derived_valu_instr_per_simd_per_cycle"] = d["SQ_INSTS_VALU"] / (1024.0 * cyc)
            d["derived_valu_issue_frac_at_2cyc"] = d["derived_valu_instr_per_simd_per_cycle"] * 2.0
        if "SQ_ACTIVE_INST_VALU" in d and d.get("SQ_INSTS_VALU", 0) > 0:
            d["derived_active_cycles_per_valu_instr"] = 4.0 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
                if c in d:
                    d["derived_frac_of_wave_cycles_" + c] = d[c] / d["SQ_WAVE_CYCLES"]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
