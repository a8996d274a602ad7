python tools/shard_scaling.py > gpurun_out/r5_shard_scaling2.txt 2>&1
python tools/fused_times.py > gpurun_out/r5_fused_times2.txt 2>&1
python tools/exp.py --check --cases ase,seed > gpurun_out/r5_record_check2.txt 2>&1
