import os, subprocess, sys
for lanes in (1, 4, 16, 64):
    env = dict(os.environ, AR_LANES_PER_WAVE=str(lanes))
    print("lanes", lanes, flush=True)
    subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); from tools.quick_gpu_timing import run; run(5,5,5,30,4096,4096,200,8)"], env=env)
