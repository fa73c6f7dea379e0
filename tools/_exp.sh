python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" 2>&1 | tail -2
for v in 1 0; do echo "== tap_inner $v"; CA_CONV_TAP_INNER=$v python tools/pp_check.py --time 2>&1 | grep conv; done
for v in 1 0; do CA_CONV_TAP_INNER=$v python bench.py --no-vae --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tap_inner $v', d['ms_per_step'])"; done
