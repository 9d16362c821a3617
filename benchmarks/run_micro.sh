python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "stream" 2>&1 | tail -2
python benchmarks/micro_stream.py 2>&1 | grep -E "fwd_stream"
echo TN2; GS_STREAM_FWD_TN=2 python benchmarks/micro_stream.py 2>&1 | grep -E "fwd_stream"
