python -m pytest tests/test_gpu_variants.py -x -q -k "layernorm" 2>&1 | tail -2
for f in 0 1 0 1; do LLA_VIT_LN_FUSE=$f python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f', d['value'], d['ms_per_step'], d['verified'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"; done
