"""Does libalpharat_hip see the GPU when torch is loaded in the same process? (order matters)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def mine():
    from alpharat_amd import _lib
    L = _lib.load()
    print("ar_device_count:", L.ar_device_count(), flush=True)
def theirs():
    import torch
    print("torch", torch.__version__, "cuda available:", torch.cuda.is_available(), "count", torch.cuda.device_count(), flush=True)
if order == "torch-first":
    theirs(); mine()
elif order == "import-only-first":
    import torch; mine(); theirs()
else:
    mine(); theirs(); mine()
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
