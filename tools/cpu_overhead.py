"""Host enqueue time per denoise step vs GPU time (is the loop ever CPU-bound?): python tools/cpu_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-vae"]
args = bench.parse()
from controlanimate_amd import kernels as K
from controlanimate_amd.controlresiduals_pipeline import MultiControlNetResidualsPipeline
device = torch.device("cuda", 0)
unet, nets = bench.build_models(args, device, torch.float16)
f, hw = args.frames, args.size // 8
g = torch.Generator().manual_seed(0)
prompt = (torch.randn(2, 77, 768, generator=g) * 0.5).to(device)
cn = MultiControlNetResidualsPipeline(["c0"], [1.0], use_lcm=False, controlnets=nets, device=device)
cn.prep_control_images([h for h in torch.rand(f, 3, args.size, args.size, generator=g)], do_classifier_free_guidance=True, guess_mode=False)
x = torch.randn(2 * f, hw, hw, unet.conv_in.cin_pad, device=device).half()
def step():
    down = cn.residuals_nhwc_async(x, 500, prompt, False)
    return unet.forward_nhwc(x, 2, f, 500, prompt, down, None)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.1f} ms/step, GPU-complete {1e3*(t2-t0)/10:.1f} ms/step")
