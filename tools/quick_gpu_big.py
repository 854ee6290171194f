import sys
sys.path.insert(0, ".")
from tools.quick_gpu_timing import run
run(5, 5, 5, 30, 131072, 131072, 200, 8)
run(5, 5, 5, 30, 262144, 262144, 200, 8)
run(7, 7, 10, 50, 65536, 65536, 400, 16, c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
