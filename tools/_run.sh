L=raytrace-miniapp_amd/csrc
timeout -k 10 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 500 python tools/exp.py --cases ase,small,shard8,seed $L/librt_hip.so $L/librt_hip_prev.so 2>&1
timeout -k 10 500 python tools/exp.py --cases ase,small,shard8,seed $L/librt_hip_prev.so $L/librt_hip.so 2>&1
