"""Where a denoise step's wall time goes that is NOT kernel execution: from a `rocprofv3 --kernel-trace` CSV of a bench.py run,
  * the union of all kernel intervals of a steady-state window of the trace (both streams): busy vs idle time of the GPU,
  * the gaps between consecutive kernels of the busier queue, by size class,
  * time during which exactly one / two (or more) kernels are in flight, and the kernels that most often run alone after a gap.
    python tools/timeline_gaps.py <kernel_trace.csv> [steps_in_window [steps_to_skip_at_the_end]]
Steps are delimited by the launches of the temporal-attention kernel of the 64x64-latent level (10 per step of BASELINE config 2).
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    marks = [s for s, e, n, q in rows if "k_tattn_out" in n or "k_tattn_fused" in n]
    per_step = 10
    nsteps = len(marks) // per_step
    # bench.py ends with an INSTRUMENTED pass (5 eager single-stream steps with an event pair around every GEMM launch) and probe
    # launches: the window must end before them -- `skip_last` steps from the end (default 7), inside the timed, graph-replayed steps
    skip_last = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    want = int(sys.argv[2]) if len(sys.argv) > 2 else min(10, nsteps - skip_last - 4)
    first = (nsteps - want - skip_last) * per_step
    t0, t1 = marks[first], marks[first + want * per_step]
    win = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in rows if e > t0 and s < t1]
    span = (t1 - t0) / 1e6
    print(f"{path}: {nsteps} steps in the trace; window of {want} steps = {span:.3f} ms ({span / want:.3f} ms per step), {len(win)} kernels ({len(win) / want:.0f} per step)")
    # sweep: number of kernels in flight over time
    ev = []
    for s, e, n, q in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, by_depth = 0, t0, defaultdict(int)
    for t, d in ev:
        by_depth[min(depth, 3)] += t - last
        last = t
        depth += d
    by_depth[min(depth, 3)] += t1 - last
    for k in sorted(by_depth):
        print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {by_depth[k] / 1e6 / want:7.3f} ms per step ({100.0 * by_depth[k] / (t1 - t0):5.1f} %)")
    # per queue: busy time and gaps
    queues = defaultdict(list)
    for s, e, n, q in win:
        queues[q].append((s, e, n))
    for q, ks in sorted(queues.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in ks)
        gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
        small = [g for g in gaps if 0 <= g[0] < 10_000]
        mid = [g for g in gaps if 10_000 <= g[0] < 100_000]
        big = [g for g in gaps if g[0] >= 100_000]
        print(f"  queue {q}: {len(ks) / want:.0f} kernels per step, busy {busy / 1e6 / want:.3f} ms per step; gaps < 10 us: {len(small) / want:.0f} per step, "
              f"{sum(g[0] for g in small) / 1e6 / want:.3f} ms (median {sorted(g[0] for g in small)[len(small) // 2] / 1e3 if small else 0:.2f} us); "
              f"10-100 us: {len(mid) / want:.1f} per step, {sum(g[0] for g in mid) / 1e6 / want:.3f} ms; >= 100 us: {len(big) / want:.1f} per step, "
              f"{sum(g[0] for g in big) / 1e6 / want:.3f} ms")
        after = defaultdict(lambda: [0, 0])
        for g, a, b in mid + big:
            after[(short(a), short(b))][0] += 1
            after[(short(a), short(b))][1] += g
        for (a, b), (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:6]:
            print(f"      {c / want:5.1f} per step, {t / 1e6 / want:.3f} ms: {a}  ->  {b}")
        # what the OTHER queues run while this one waits (its first gap >= 100 us that is not the end of a step: < 10 ms)
        waits = [(ks[i][1], ks[i + 1][0]) for i in range(len(ks) - 1) if 100_000 <= ks[i + 1][0] - ks[i][1] < 10_000_000]
        if waits:
            g0, g1 = waits[len(waits) // 2]
            print(f"      during one such wait of {(g1 - g0) / 1e3:.0f} us the other queues run:")
            for s2, e2, n2, q2 in win:
                if q2 != q and e2 > g0 and s2 < g1:
                    print(f"        +{(s2 - g0) / 1e3:8.1f} us  {(e2 - s2) / 1e3:7.1f} us  {short(n2)}")


if __name__ == "__main__":
    main()
