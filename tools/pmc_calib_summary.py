"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of tools/probe_pmc_calib (1 GiB per kernel) -> bytes per counter unit.
    python tools/pmc_calib_summary.py fetch_counter_collection.csv write_counter_collection.csv > profiles/round5_pmc_calibration.txt"""
import csv, re, sys

GIB = float(1 << 30)


def read(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = re.sub(r"^void ", "", r["Kernel_Name"])
            k = k[:k.index(">(") + 1] if ">(" in k else k.split("(")[0]  # (template arguments may hold parentheses: __vector(4))
            out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return out


f, w = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
print("# tools/probe_pmc_calib.hip under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); every kernel moves 1 GiB = 1048576 KB")
print("# factor = true bytes / (counter x 1024): what a counter value must be multiplied by in this access shape")
print(f"{'kernel':52s} {'FETCH_SIZE KB':>14s} {'factor':>7s} {'WRITE_SIZE KB':>14s} {'factor':>7s}")
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_calib"):
        continue
    fv, wv = f.get(k, 0.0), w.get(k, 0.0)
    is_read = "read" in k
    ff = f"{GIB / (fv * 1024):7.3f}" if (is_read and fv > 0) else "      -"
    wf = f"{GIB / (wv * 1024):7.3f}" if (not is_read and wv > 0) else "      -"
    print(f"{k:52s} {fv:14.0f} {ff} {wv:14.0f} {wf}")
