"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_summary.py file.csv COUNTER [COUNTER...]"""
import collections, csv, re, sys
path, counters = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if "k_" not in k: continue
    k = re.sub(r"\(anonymous namespace\)::", "", k)[:56]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == counters[0]: n[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][counters[0]])[:14]:
    print(f"{k:58s} launches={n[k]:5d} " + " ".join(f"{c}={v[c]:.4g} (per launch {v[c]/max(n[k],1):.4g})" for c in counters))
