python tools/env_ab.py seed --rounds 2 --envs "-;RT_HIP_MARCH_MODE=2;RT_HIP_MARCH_MODE=3;RT_HIP_MARCH_MODE=4" 2>&1 | cut -c1-110
