"""Which thread of a HIP process burns a core while the GPU is busy, and does any runtime setting stop it?
A hipGraph of small kernels is replayed continuously, paced as ControlAnimationPipeline paces its steps (hipEventQuery polls between
naps, two replays in flight); the script reports CPU time per thread and, with --bt, the backtrace of the hottest non-main thread.
    python tools/host_spin_probe.py [--bt] [--eager]        (environment variables under test are set by the caller)"""
import ctypes, os, subprocess, sys, threading, time

import torch


def thread_cpu():
    out, tick = {}, os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as fh:
                rest = fh.read().rsplit(")", 1)[1].split()
            out[int(tid)] = (int(rest[11]) + int(rest[12])) / tick
        except OSError:
            pass
    return out


def main():
    bt, eager = "--bt" in sys.argv, "--eager" in sys.argv
    a = torch.randn(2048, 2048, device="cuda", dtype=torch.float16)
    b = torch.randn(2048, 2048, device="cuda", dtype=torch.float16)
    c = torch.empty_like(a)

    def work():
        for _ in range(400):
            torch.mm(a, b, out=c)

    work()
    torch.cuda.synchronize()
    g = None
    if not eager:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            work()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); (g.replay() if g else work()); e1.record(); torch.cuda.synchronize()
    replay_ms = e0.elapsed_time(e1)
    lib = None
    if bt:
        so = "/tmp/thread_bt.so"
        subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(os.path.abspath(__file__)), "thread_bt.c"), "-ldl"], check=True)
        lib = ctypes.CDLL(so)
        lib.ca_bt_install()
    t0, c0, w0 = thread_cpu(), time.process_time(), time.perf_counter()
    ring, n = [], 0
    while time.perf_counter() - w0 < 3.0:
        if len(ring) >= 2:
            ev = ring.pop(0)
            while not ev.query():
                time.sleep(0.001)
        (g.replay() if g else work())
        ev = torch.cuda.Event()
        ev.record()
        ring.append(ev)
        n += 1
        if bt and n in (10, 20, 30):
            t1 = thread_cpu()
            hot = max((t for t in t1 if t != os.getpid()), key=lambda t: t1[t] - t0.get(t, 0.0))
            lib.ca_bt_signal(hot)
            time.sleep(0.05)
    for ev in ring:
        while not ev.query():
            time.sleep(0.001)
    wall, cpu = time.perf_counter() - w0, time.process_time() - c0
    t1 = thread_cpu()
    top = sorted(((t1[t] - t0.get(t, 0.0), t) for t in t1), reverse=True)[:3]
    tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.split("_")[0] in ("HSA", "ROC", "DEBUG", "AMD", "GPU") and k not in ("HSA_XNACK", "HSA_ENABLE_IPC_MODE_LEGACY", "ROCR_VISIBLE_DEVICES", "ROCM_PATH"))
    print(f"[{tag or 'default'}]{' eager' if eager else ''} replay {replay_ms:.1f} ms x {n} in {wall:.2f} s wall; process CPU {cpu / wall * 100:.0f}% of one core; "
          f"threads (main = {os.getpid()}): " + ", ".join(f"{t}{'(main)' if t == os.getpid() else ''}: {d / wall * 100:.0f}%" for d, t in top), flush=True)


main()
