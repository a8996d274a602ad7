python tools/fused_times.py > gpurun_out/r5_fused_times_def.txt 2>&1
python tools/wave_trace.py shard8 2>&1 | grep -v kcycles > gpurun_out/r5_trace_shard8_def.txt
