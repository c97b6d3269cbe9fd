for r in 1 2; do python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d['config']['tower_batch'])"; done
python bench.py --dataset-images 1000000 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-260
python -m pytest tests/test_gpu_compressor.py tests/test_gpu_bench_config.py -x -q 2>&1 | tail -2
