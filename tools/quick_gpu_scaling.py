import sys
sys.path.insert(0, ".")
from tools.quick_gpu_timing import run
for conc in (4096, 16384, 65536):
    run(5, 5, 5, 30, conc, conc, 200, 8)
run(7, 7, 10, 50, 16384, 16384, 400, 16, c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
