#!/usr/bin/env python3
"""profiles/valu.json (what bench.py quotes as `roofline_valu`) from an issue-counter summary of tools/profile_issue_counters.sh.
usage: make_valu_json.py <issue_counters.json> <build tag> [transforms in the probe = 4] > profiles/valu.json
Vector instructions per pixel of one 8192^2 transform, by kernel, and the cycles a SIMD spends on one of them (the SQ's own
count: SQ_ACTIVE_INST_VALU over SQ_INSTS_VALU, quad-cycles x 4)."""
import json, sys
src, tag = sys.argv[1], sys.argv[2]
transforms = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # pmc_probe.py --steps 3 + 1 calibration transform
d = json.load(open(src))
d = d.get("kernels", d)
npx = 8192 * 8192
per, tot_instr, tot_active = {}, 0.0, 0.0
for name, v in d.items():
    short = name.replace("void ", "").replace("wsk::", "")
    if not (short.startswith("k_relax<") or short.startswith("k_resolve_local<true, false") or short in ("k_resolve_chase", "k_seed_tables")):
        continue
    lp = v["launches"] / transforms
    instr = v["SQ_INSTS_VALU"]                        # wave-instructions per launch
    active = 4.0 * v["SQ_ACTIVE_INST_VALU"]           # SIMD cycles with a vector instruction active, per launch
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0                  # kernel cycles (the counter sums the 8 XCDs)
    per[short] = {"launches_per_transform": lp, "wave_instr_per_launch": instr, "lane_instr_per_px_of_the_plane": instr * 64.0 / npx * lp,
                  "cycles_per_wave_instr": active / instr if instr else None, "valu_active_frac_of_simd_cycles": active / (1024.0 * cyc) if cyc else None}
    tot_instr += instr * lp
    tot_active += active * lp
out = {
    "build": tag,
    "source": f"profiles/{tag}_issue_counters.json: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE ... on tools/pmc_probe.py (8192x8192, "
              "3 transforms + 1 calibration transform; tools/profile_issue_counters.sh), SINGLE-CONTEXT profile; SQ cycle counters are quad-cycles",
    "lane_instr_per_px": tot_instr * 64.0 / npx,
    "wave_instr_per_transform": tot_instr,
    "simd_cycles_valu_active_per_transform": tot_active,
    "cycles_per_wave_instr_measured": tot_active / tot_instr,
    "microbenchmark": "profiles/r3_v0_valu_microbench.txt + r3_v0_valu_ops.txt (tools/microbench_valu*.hip): v_add_u32 / v_and / v_mov 3.2 cycles per wave-instruction and SIMD, "
                      "v_min_u32 / v_min3_u32 / v_med3_u32 / every VOP3 and packed op 4.5-4.9, DPP moves 5.8; the relaxation's pixel update (min3, min, add, med3) 17.4 cycles = "
                      "8.4 T pixel updates/s for the chip, whatever the occupancy (1-8 waves per SIMD)",
    "per_kernel": per,
}
print(json.dumps(out, indent=1))
