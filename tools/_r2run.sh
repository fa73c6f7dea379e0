R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2_pytest_gpu.txt
python bench.py > gpurun_out/round2_bench.json 2> gpurun_out/r2_bench.err
for c in 1 3 4 5; do python bench.py --config $c --no-vae --no-cpu-baseline > gpurun_out/round2_bench_config$c.json 2> gpurun_out/r2_bench_c$c.err; done
CA_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 1 --no-vae --no-cpu-baseline --no-roofline > gpurun_out/round2_bench_2ranks_one_gpu.json 2> gpurun_out/r2_bench_2r.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof -o r2 -- python3 $R/bench.py --steps 4 --warmup 2 --no-graph --no-overlap --no-cpu-baseline --no-vae > $R/gpurun_out/r2_prof_bench.json 2> $R/gpurun_out/r2_prof.err
rm -f $R/gpurun_out/r2_prof/*kernel_trace.csv $R/gpurun_out/r2_prof/*.db
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-overlap --no-graph > /dev/null 2> $R/gpurun_out/r2_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-roofline --no-overlap --no-graph > /dev/null 2> $R/gpurun_out/r2_pmc_w.err
cd $R
python tools/pmc_traffic_json.py gpurun_out/r2_pmc_f/f_counter_collection.csv gpurun_out/r2_pmc_w/w_counter_collection.csv config2 fp16 3 > gpurun_out/round2_pmc_traffic.json
rm -rf gpurun_out/r2_pmc_f gpurun_out/r2_pmc_w
