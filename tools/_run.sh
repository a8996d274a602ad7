python tools/env_ab.py seed --rounds 2 --envs "-;RT_HIP_LATE2_X10=10;RT_HIP_LATE2_X10=20;RT_HIP_LATE2_X10=32;RT_HIP_LATE2_X10=60;RT_HIP_LATE2_X10=32,RT_HIP_LATE_WAVES=8" > gpurun_out/r5_late_twokernel.txt 2>&1
export RT_HIP_FUSED=2
python tools/env_ab.py standin shard8 small --rounds 1 --envs "-;RT_HIP_LATE2_X10=20;RT_HIP_LATE2_X10=32;RT_HIP_LATE2_X10=60" >> gpurun_out/r5_late_twokernel.txt 2>&1
