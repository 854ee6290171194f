import ctypes as C, sys, shutil
sys.path.insert(0, ".")
from alpharat_amd import _lib
_lib.LIB_PATH = _lib.PKG / "libalpharat_hip_stats.so"
from alpharat_amd.sampling import rust_self_play
L = _lib.load()
names_g = ["PICK", "ALLOC", "CHILD", "ENTER"]; names_b = ["ENTRY", "LEVEL", "CANCEL", "CANCEL_LEVEL"]
def dump(tag):
    out = (C.c_ulonglong * 32)()
    L.ar_debug_round_stats.argtypes = [C.c_void_p]
    L.ar_debug_round_stats(out)
    o = list(out)
    tot = o[31]
    print(tag, "rounds", tot, "avg alive lanes %.1f" % (o[30] / max(tot, 1)))
    for mi, names in ((0, names_g), (1, names_b)):
        for si, nm in enumerate(names):
            r, l = o[mi * 12 + si * 2], o[mi * 12 + si * 2 + 1]
            if r: print("   %-14s rounds %10d (%.1f%%)  lanes/round %.1f" % (("G_" if mi == 0 else "B_") + nm, r, 100.0 * r / tot, l / r))
rust_self_play(width=5, height=5, cheese_count=5, max_turns=30, num_games=16384, simulations=200, batch_size=8, output_dir=None, seed=0, concurrent_games=16384)
dump("5x5 uniform fused")
rust_self_play(width=7, height=7, cheese_count=10, max_turns=50, num_games=8192, simulations=400, batch_size=16, output_dir=None, seed=0, concurrent_games=8192, c_puct=0.512, fpu_reduction=0.459, force_k=0.103, noise_epsilon=0.25)
dump("7x7 uniform fused")
