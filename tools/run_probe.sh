export LLA_VIT_STREAMS=2
for v in "" dma_rmw_sc0 dma_sc0 "" dma_rmw_sc0 dma_sc0; do
  if [ -n "$v" ]; then export LLA_LIB=$PWD/lossyless_amd/variants/liblossyless_amd_$v.so; else unset LLA_LIB; fi
  echo "== two lanes, deferred, variant ${v:-default}"; python tools/pipeline_probe.py 4000000 same 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
done
