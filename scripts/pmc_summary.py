"""Average each PMC counter per kernel name from a rocprofv3 counter_collection CSV."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("ws::(anonymous namespace)::", "").replace("void ", "")[:44]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(ctrs.items())]
    n = max(len(v) for v in ctrs.values())
    print(f"{name:46s} n={n:4d}  " + "  ".join(parts))
