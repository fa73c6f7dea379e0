"""rocprofv3 --pmc SQ_* counter CSV -> per-kernel percentages of wave cycles (profiles/roundN_pmc_sq_summary.txt).
    python tools/pmc_sq_summary.py counter_collection.csv > profiles/round2_pmc_sq_summary.txt
Counters expected in ONE pass: SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"""
import collections, csv, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0]
    if not k.startswith("k_"): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
print("# share of SQ_WAVE_CYCLES per kernel (all launches of one bench.py pass: --steps 2 --warmup 1 --no-graph --no-overlap)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    w = v["SQ_WAVE_CYCLES"] or 1.0
    print(f"{k[:56]:56s} n={n[k]:5d} " + " ".join(f"{c[3:]}={100 * v[c] / w:.0f}%" for c in sorted(v) if c != "SQ_WAVE_CYCLES"))
