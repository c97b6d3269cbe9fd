python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/b3.json 2> gpurun_out/b3.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','verified')}, "gemm", d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['verification'], flush=True)
print(d['stl10_shaped_stage'], d['rn50_stage'])
PY
echo "== default pipeline 6M"; python tools/pipeline_probe.py 6000000 same 2>&1 | grep -v amdgpu | tail -1 | cut -c1-250
