"""Shader-clock stamps of the fused feed-forward (experiments build): where a round of block 0's first tile spends its cycles.
    CA_HIP_LIB=.../libcontrolanimate_hip_exp.so python tools/ff_stamps.py
tags: 0 tile start, 1 before B1, 2 after B1, 3 before B2, 4 after B2 (producer: 1 = K loop done, 3 = epilogue done; consumer: 3 = stage 2 done)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import ff_check
from controlanimate_amd import kernels as K

d = ff_check.make(131072, torch.float16)
for _ in range(3):
    ff_check.fused(d)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 512)()
lib = K.lib()
lib.ca_debug_ff_stamps.restype = C.c_int
assert lib.ca_debug_ff_stamps(buf) == 0
for role, name in ((0, "producer wave 0"), (1, "consumer wave 4")):
    st = [(buf[role * 256 + 2 * i], buf[role * 256 + 2 * i + 1]) for i in range(120) if buf[role * 256 + 2 * i]]
    print(name, "stamps", len(st))
    prev = st[0][0]
    line = []
    for t, tag in st:
        line.append(f"{tag}:{t - prev}")
        prev = t
        if tag == 4:
            print("  ", " ".join(line))
            line = []
    if line:
        print("  ", " ".join(line))
    print("   total", st[-1][0] - st[0][0])
