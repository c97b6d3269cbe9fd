python -m pytest tests/test_gpu_compressor.py tests/test_gpu_distributed.py tests/test_preprocess.py -x -q 2>&1 | tail -2
python bench.py --dataset-images 1000000 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-250
python bench.py --host-images 40960 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | cut -c1-300
