import sys
sys.path.insert(0, ".")
from tools.quick_gpu_timing import run
run(5, 5, 5, 30, 4096, 4096, 200, 8)
