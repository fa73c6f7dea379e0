"""rocprofv3 --pmc MFMA counters -> per-kernel MFMA utilisation (profiles/roundN_mfma_util.txt).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d DIR -o run -- \
        python3 bench.py --steps 2 --warmup 1 --no-graph --no-overlap --no-roofline --no-vae --no-cpu-baseline
    python tools/pmc_mfma_summary.py DIR/.../run_counter_collection.csv > profiles/round3_mfma_util.txt

Definitions (MI355X_MICROARCH.md, "Per-instruction cycle constants"):
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, summed over all SIMDs of the chip
    (= 16 x the number of v_mfma_f32_16x16x32 instructions: checked in round 1);
  * GRBM_GUI_ACTIVE = GPU-active cycles of the dispatch, SUMMED OVER THE 8 XCDs by rocprofv3 (checked: for k_gemm_ps the
    counter per launch is 1.68 M = 8 x 210 k cycles = 8 x 87.5 us x 2.4 GHz, the launch duration of the kernel trace); the
    chip has 256 CUs x 4 SIMDs = 1024 matrix pipes, so
        mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024);
  * SQ_INSTS_VALU_MFMA_MOPS_F16 counts fp16 MFMA work in units of 512 FLOP (rocprof's MfmaFlopsF16 = MOPS x 512):
        executed TFLOP (incl. tile padding) per launch, and with the dispatch's active cycles a clock-independent FLOP/cycle.
"""
import collections
import csv
import re
import sys

SIMDS = 1024
XCDS = 8
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"^void ", "", k).split("(")[0]
    if not k.startswith("k_"):
        continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key)
        launches[k] += 1
print("# per kernel, all launches of one bench.py pass (see the module docstring for the command and the definitions)")
print(f"# {'kernel':58s} {'launches':>8s} {'mfma_util':>9s} {'FLOP/clk/SIMD':>13s} {'exec TFLOP/launch':>17s} {'busy share of step':>18s}")
step_active = sum(v.get("GRBM_GUI_ACTIVE", 0.0) / XCDS for v in tot.values()) or 1.0
wm = wa = 0.0
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
    act = v.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mops = v.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    if act <= 0:
        continue
    util = busy / (act * SIMDS)
    wm += busy
    wa += act
    print(f"{k[:60]:60s} {launches[k]:8d} {100 * util:8.1f}% {mops * 512 / (act * SIMDS):13.1f} {mops * 512 / max(launches[k], 1) / 1e12:17.4f} {100 * act / step_active:17.1f}%")
print(f"# all k_* kernels: mfma_util = {100 * wm / (wa * SIMDS):.1f}% of the matrix pipes' cycles (peak = 1017 FLOP/clk/SIMD for fp16 16x16x32)")
