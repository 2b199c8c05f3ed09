#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KiB
(bytes = value * 1024); on gfx950 FETCH_SIZE under-reports coalesced streaming reads, exactly 1/2 for
16 B/lane loads and uncalibrated for other widths, so the read side is calibrated on a copy kernel of
known size that runs in the same profile (tools/pmc_probe.py: k_flood_step reads 5 B and writes 4 B
per pixel with the engine's own 4 B/lane row accesses).
Usage: pmc_summarise.py <fetch_counter_csv> <write_counter_csv> <pixels>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(set)
    big = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        tot[name] += float(row["Counter_Value"])
        cnt[name].add(row["Dispatch_Id"])
        big[name] = max(big[name], float(row["Counter_Value"]))
    return {k: (tot[k], len(cnt[k]), big[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    n_words = int(sys.argv[3])
    cal_name = next((k for k in fetch if "k_flood_step" in k), None)
    scale4 = scale16 = 1.0
    cal = {}
    if cal_name:
        raw = fetch[cal_name][0] / fetch[cal_name][1] * 1024
        scale4 = (5.0 * n_words) / raw if raw else 1.0
        wraw = write.get(cal_name, (0, 1, 0))
        cal["4B_per_lane"] = {"kernel": cal_name, "true_read_bytes": 5 * n_words, "FETCH_SIZE_bytes_raw": raw, "read_scale": scale4,
                              "WRITE_SIZE_bytes_raw": wraw[0] / max(wraw[1], 1) * 1024, "true_write_bytes": 4 * n_words}
    copy_name = next((k for k in fetch if "copyBuffer" in k), None)
    if copy_name and fetch[copy_name][2] * 1024 > n_words:      # the one plane-sized copy (largest dispatch)
        raw = fetch[copy_name][2] * 1024
        scale16 = (4.0 * n_words) / raw
        cal["16B_per_lane"] = {"kernel": copy_name + " (largest dispatch)", "true_read_bytes": 4 * n_words,
                               "FETCH_SIZE_bytes_raw": raw, "read_scale": scale16,
                               "WRITE_SIZE_bytes_raw": write.get(copy_name, (0, 1, 0))[2] * 1024, "true_write_bytes": 4 * n_words}
    out = {"calibration": cal, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, fn, _ = fetch.get(k, (0.0, 0, 0.0))
        w, wn, _ = write.get(k, (0.0, 0, 0.0))
        out["kernels"][k] = {
            "launches": max(fn, wn),
            "fetch_bytes_per_launch_raw": f / max(fn, 1) * 1024,
            "fetch_bytes_per_launch_scaled_4B": f / max(fn, 1) * 1024 * scale4,
            "fetch_bytes_per_launch_scaled_16B": f / max(fn, 1) * 1024 * scale16,
            "write_bytes_per_launch": w / max(wn, 1) * 1024,
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
