"""Builds profiles/sq_counters.json from the four SQ counter passes of scripts/collect_profiles.sh (gpurun_out/sq/p1..p4:
one 1920x1080xF chunk, CDF 9/7 q=80, through scripts/profile_run.py F; pass F, default 64).  Counters are sums over the 8 XCDs and are
averaged over the dispatches of a kernel.

valu_issue_frac: share of the VALU issue capacity the kernel used, with the rates scripts/probes/valu_rate_probe.hip
measured on this chip (profiles/r02_valu_rate_probe.log): an add / shift / SDWA / bit-field instruction takes 2 cycles of a
SIMD per wave, a multiply (v_mad_i32_i24, v_mul_hi_u32, v_mul_lo_u32), v_perm_b32 and the packed / dot instructions 4.
The instruction mix comes from the kernel's ISA (share of 4-cycle instructions among its VALU instructions, counted
statically); cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the XCDs)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
px = 1920 * 1080 * frames
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in range(1, 5):
    files = sorted(glob.glob(os.path.join(ROOT, f"gpurun_out/sq/p{p}/*/*_counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "alice" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))

# static share of 4-cycle VALU instructions per kernel, from the device ISA
slow = re.compile(r"^\s+v_(mad_i32_i24|mad_u32_u24|mul_hi_u32|mul_lo_u32|mul_i32_i24|mul_u32_u24|perm_b32|pk_|dot|mad_i64_i32|mad_u64_u32|mul_hi_i32)")
anyv = re.compile(r"^\s+v_")
mix = {}
try:
    for src in ("transform", "rans", "generic"):
        asm = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-",
                              os.path.join(ROOT, "alice-codec_amd", "csrc", src + ".hip")], capture_output=True, text=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^(_ZN5alice\w+):", line)
            if m:
                cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
                mix[cur] = [0, 0]
            elif cur and anyv.match(line):
                mix[cur][0] += 1
                mix[cur][1] += 1 if slow.match(line) else 0
except Exception as e:  # noqa: BLE001
    print("ISA mix unavailable:", e, file=sys.stderr)

out = {"source": __doc__.strip(), "pixels": px, "per_kernel": {}}
for k, c in sorted(acc.items()):
    d = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    share4 = (mix[k][1] / mix[k][0]) if k in mix and mix[k][0] else None
    d["valu_lane_ops_per_pixel"] = round(d.get("SQ_INSTS_VALU", 0.0) * 64 / px, 2)
    if cyc > 0 and share4 is not None:
        simd_cycles = d.get("SQ_INSTS_VALU", 0.0) * (2.0 * (1 - share4) + 4.0 * share4)
        d["static_share_of_4_cycle_valu_instructions"] = round(share4, 3)
        d["valu_issue_frac"] = round(simd_cycles / (1024.0 * cyc), 3)
    if cyc > 0:
        d["lds_busy"] = round(d.get("SQ_LDS_IDX_ACTIVE", 0.0) / (256.0 * cyc), 3)
        d["lds_bank_conflict_share"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 3)
    out["per_kernel"][k] = {n: (round(v, 3 if abs(v) < 100 else 1) if isinstance(v, float) else v) for n, v in d.items()}
json.dump(out, open(os.path.join(ROOT, "profiles", "sq_counters.json"), "w"), indent=1)
for k, d in out["per_kernel"].items():
    if "xy" in k or "_t_" in k:
        print(k, {n: d.get(n) for n in ("valu_lane_ops_per_pixel", "valu_issue_frac", "lds_busy", "lds_bank_conflict_share")})
