"""Do the fp64 matrix pipe and the fp64 vector pipe of gfx950 run concurrently? (development probe)"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from accbpg_and_fw_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda")
iters = 20000
for mode, name in [(1, "8 MFMA f64 per trip"), (2, "128 v_fma_f64 per trip"), (3, "both")]:
    ms = C.c_double(0)
    assert lib.accbpg_debug_pipe_probe(iters, mode, C.byref(ms), None) == 0
    mf = 256 * 4 * iters * 8 * 2048 if mode & 1 else 0
    vf = 256 * 4 * iters * 128 * 64 * 2 if mode & 2 else 0
    print("%-24s %8.3f ms   matrix %.1f TFLOP/s  vector %.1f TFLOP/s" % (name, ms.value, mf / ms.value / 1e9, vf / ms.value / 1e9))
