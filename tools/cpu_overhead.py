"""Host time per denoise step of the PRODUCT loop (ControlAnimationPipeline.__call__, hipGraph default) vs the GPU time: is the loop ever
CPU-bound?  `bench.py` measures it in its timed region (`host_enqueue_ms_per_step`: the calls have returned, the device is still
working); this prints that figure with the graph and without it:  python tools/cpu_overhead.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in ([], ["--no-graph"]):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--no-cpu-baseline", "--no-cpu-baseline-config2", "--no-roofline", "--no-vae"] + extra,
                         capture_output=True, text=True, check=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(f"{'hipGraph' if not extra else 'eager   '}: host CPU {d['host_cpu_ms_per_step']:.2f} ms/step, calls returned after {d['host_enqueue_ms_per_step']:.2f} ms/step, GPU-complete {d['ms_per_step']:.2f} ms/step")
