"""Builds profiles/traffic.json from the two PMC passes written by scripts/collect_profiles.sh
(gpurun_out/final/pmc_fetch, gpurun_out/final/pmc_write; pass the frame count of the profiled chunk): HBM bytes per pixel and per launch of every kernel.

The unit of the two counters as rocprofv3 reports them is found from a kernel with known traffic (inv_t_kernel reads
exactly 3 bytes of symbols per pixel): it comes out as 1 KB.  On gfx950 FETCH_SIZE reports exactly half of the bytes read, for 4, 8 and 16 bytes per
lane and for the strided segment pattern of the tile kernels (scripts/probes/traffic_calib.hip and the calibration
copies at the end of scripts/profile_run.py), so it is doubled; WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
px = 1920 * 1080 * frames


def per_kernel(pattern, counter):
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(ROOT, pattern)), key=os.path.getmtime)
    # newest run only (gpurun_out accumulates)
    newest = files[-1:]
    for f in newest:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


fetch = per_kernel("gpurun_out/final/pmc_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
write = per_kernel("gpurun_out/final/pmc_write/*/*_counter_collection.csv", "WRITE_SIZE")
# calibration: the copy kernels at the end of profile_run.py move 256 MiB each way
cal = {}
for k, v in fetch.items():
    if "copy" in k.lower() or "elementwise" in k.lower():
        cal[k] = [x for x in v]
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/collect_profiles.sh) on "
                 f"scripts/profile_run.py {frames} (one 1920x1080x{frames} chunk, CDF 9/7 q=80); counter unit calibrated on inv_t_kernel's known reads; "
                 "FETCH_SIZE doubled (gfx950 reports exactly half of known reads: scripts/probes/traffic_calib.hip), WRITE_SIZE as is; "
                 "the chunk the bench times when run with 64",
       "per_kernel": {}}
unit = None
for k in sorted(set(fetch) | set(write)):
    if "alice" not in k:
        continue
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    out["per_kernel"][k] = {"fetch_counter": f, "write_counter": w}
# the unit is fixed by the known traffic of inv_t (reads 3 B/px of symbols): pick 1, 32 or 64 bytes per count
ref = out["per_kernel"].get(next((k for k in out["per_kernel"] if "inv_t_kernel" in k), ""), None)
scale = 64.0
if ref and ref["fetch_counter"] > 0:
    for cand in (1.0, 32.0, 64.0, 1024.0):
        if 0.5 < ref["fetch_counter"] * cand * 2 / (3.0 * px) < 2.0:
            scale = cand
out["counter_unit_bytes"] = scale
fw = iv = 0.0
for k, d in out["per_kernel"].items():
    d["fetch_bytes_per_pixel"] = round(d.pop("fetch_counter") * scale * 2 / px, 3)
    d["write_bytes_per_pixel"] = round(d.pop("write_counter") * scale / px, 3)
    tot = d["fetch_bytes_per_pixel"] + d["write_bytes_per_pixel"]
    if "fwd_" in k:
        fw += tot
    if "inv_" in k:
        iv += tot
out["forward_transform_bytes_per_pixel"] = round(fw, 3)
out["inverse_transform_bytes_per_pixel"] = round(iv, 3)
out["forward_transform_hbm_bytes_per_launch"] = int(fw * 1920 * 1080 * 64)
out["inverse_transform_hbm_bytes_per_launch"] = int(iv * 1920 * 1080 * 64)
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k != "per_kernel" and k != "source"}))
