"""Sum rocprofv3 --pmc counter_collection CSV rows per kernel name: tools/pmc_summary.py file.csv [...]"""
import collections
import csv
import sys



def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("prb::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):  # cut the parameter list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return name[:90]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for path in sys.argv[1:]:
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            name = short(row["Kernel_Name"])
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row["Dispatch_Id"], path)
            if key not in seen:
                seen.add(key)
                calls[name] += 1
for name, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(name, "dispatches", calls[name])
    for k, v in sorted(c.items()):
        print(f"    {k:28s} {v:.4g}")
