for g in 1 0 1 0; do LLA_TOWER_GATHER=$g python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gather=$g', d['value'], d['ms_per_step'], d['roofline']['frac'], d['verified'], d['verification']['records_equal_oracle'])"; done
