"""usage: python tools/timeline.py <stats_kernel_trace.csv> [pattern] [last_n] -- start / duration of the last dispatches whose name matches"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "bcfgpu"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sel = [r for r in rows if any(p in r["Kernel_Name"] for p in pat.split(","))]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in sel[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    name = r["Kernel_Name"].replace("bcfgpu::", "").replace("void ", "").split("(")[0]
    print("%-44s grid %8s lds %7s vgpr %4s  start %9.3f  dur %8.3f ms" % (name[-44:], r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], (s - t0) / 1e6, (e - s) / 1e6))
