#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py quotes as `roofline.traffic*`) from a pmc_hbm_bytes.json of tools/profile_round.sh.
usage: make_traffic_json.py <pmc_hbm_bytes.json> <build tag> [transforms in the probe = 4] > profiles/traffic.json"""
import json, sys
src, tag = sys.argv[1], sys.argv[2]
transforms = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # pmc_probe.py --steps 3 + 1 calibration transform
d = json.load(open(src))
npx, nseeds = 8192 * 8192, None
per, relax_f, relax_w, relax_n, tot_f, tot_w = {}, 0.0, 0.0, 0, 0.0, 0.0
for name, v in d["kernels"].items():
    short = name.replace("void ", "").replace("wsk::", "")
    if not (short.startswith("k_relax<") or short.startswith("k_resolve_local<true, false") or short in ("k_resolve_chase", "k_seed_tables")):
        continue
    lp = v["launches"] / transforms
    f, w = v["fetch_bytes_per_launch_scaled_16B"], v["write_bytes_per_launch"]
    per[short] = {"launches_per_transform": lp, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w}
    tot_f += f * lp; tot_w += w * lp
    if short.startswith("k_relax<"):
        relax_f += f * lp; relax_w += w * lp; relax_n += lp
out = {
    "build": tag,
    "source": f"profiles/{tag}_pmc_hbm_bytes.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/pmc_probe.py "
              "(8192x8192, 3 transforms + 1 calibration transform; tools/profile_round.sh); counters in KiB; read side x2.000, the gfx950 "
              f"correction calibrated in the same run on a plane-sized 16 B/lane copy (x{d['calibration']['16B_per_lane']['read_scale']:.4f}) and on "
              f"k_flood_step4 (4 B/lane: x{d['calibration']['4B_per_lane']['read_scale']:.3f}); WRITE_SIZE exact; k_relax = all variants averaged over "
              f"their {relax_n:g} launches per transform",
    "k_relax": {"launches_per_transform": relax_n, "fetch_bytes_per_launch": relax_f / relax_n, "write_bytes_per_launch": relax_w / relax_n,
                "bytes_per_launch": (relax_f + relax_w) / relax_n},
    "transform_total_bytes": tot_f + tot_w, "transform_fetch_bytes": tot_f, "transform_write_bytes": tot_w,
    "compulsory_bytes": 721103936,
    "per_kernel": per,
}
print(json.dumps(out, indent=1))
