L=raytrace-miniapp_amd/csrc
timeout -k 10 500 python tools/exp.py --cases ase,seed,small $L/librt_hip.so $L/librt_hip_prev.so 2>&1
RT_HIP_FUSED=2 timeout -k 10 500 python tools/exp.py --cases ase,small $L/librt_hip.so $L/librt_hip_prev.so 2>&1
