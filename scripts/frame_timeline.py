"""Print the kernel timeline of one steady-state frame from a rocprofv3 kernel-trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_preprocess" in r["Kernel_Name"]]
k = min(20, len(idx) - 2)
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a - 1:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("ws::(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  grid {r['Grid_Size_X']:>8}  {name[:48]}")
print(f"frame period: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
