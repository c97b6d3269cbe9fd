echo "== two lanes (opt-in), deferred, 24M"; LLA_VIT_STREAMS=2 python tools/pipeline_probe.py 24000000 same 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
echo "== two lanes, joined mode, 4000 passes"; LLA_VIT_STREAMS=2 python tools/determinism_probe.py tower 4000 2>&1 | grep -v amdgpu | tail -2 | cut -c1-200
