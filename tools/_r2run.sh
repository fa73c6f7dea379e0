python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2_pytest_gpu.txt
python bench.py --no-vae --no-cpu-baseline > gpurun_out/r2_bench_wres.json 2> gpurun_out/r2_bench_wres.err
CA_GEMM_WRES=0 python bench.py --no-vae --no-cpu-baseline > gpurun_out/r2_bench_nowres.json 2> gpurun_out/r2_bench_nowres.err
