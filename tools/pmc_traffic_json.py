"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes of the same bench.py command) -> the per-kernel
summary bench.py reads (profiles/roundN_pmc_traffic.json).
    python tools/pmc_traffic_json.py fetch_counter_collection.csv write_counter_collection.csv config2 fp16 > profiles/round2_pmc_traffic.json
gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM section) -> doubled; both counters are in KB."""
import collections, csv, json, re, sys

def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k).split("(")[0]
        if not k.startswith("k_"):
            continue
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n

f, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
w, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
kern = {}
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0.0))):
    fa, wa = f[k] / max(nf[k], 1), w.get(k, 0.0) / max(nw.get(k, 0), 1)
    kern[k] = {"launches": nf[k], "fetch_size_kb_avg": round(fa, 1), "write_size_kb_avg": round(wa, 1),
               "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
# denoise steps executed in the pass = launches of the sampler kernel (one per step: the pipeline-driven bench runs a whole
# first window before its timed region); argv[5] overrides
steps = int(sys.argv[5]) if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else (nf.get("k_cfg_scheduler_step", 0) or 3)
total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in kern.values())
print(json.dumps({
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-overlap --no-graph",
    "units": "FETCH_SIZE / WRITE_SIZE are reported in KB; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md 'HBM'; Infinity-Cache hits are included, so this is L2-miss traffic, an upper bound of HBM bytes)",
    "workload": {"key": sys.argv[3], "dtype": sys.argv[4]},
    "steps_in_the_pass": steps,
    "sum_over_kernels_bytes_per_step": int(total / steps),
    "kernels": kern}, indent=1))
