"""What a launch of the Gram kernel is made of (development aid): timing ablations of the production loop and of the
dealt-out schedule through accbpg_debug_gram_variant (wrong results except the first two), interleaved rounds."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODES = {10: "block schedule (production in round 2)", 19: "dealt-out schedule B", 20: "B, no loads in the loop",
         21: "B, no wait + barrier", 22: "B, no fragment reads", 23: "B, barrier without the vmcnt wait",
         24: "B, MFMA only", 25: "B, loads only", 26: "B, barrier only", 27: "B, reads only"}


def main():
    import torch
    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd import _lib
    from accbpg_and_fw_amd.functions import _ptr
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(3)
    V = torch.randn(2048, 32768, dtype=torch.float64, device="cuda", generator=gen)
    x = torch.rand(32768, dtype=torch.float64, device="cuda", generator=gen) + 0.05
    x /= x.sum()
    f = acc.DOptimalObj(V)
    ms = C.c_double(0.0)
    out = {c: [] for c in CODES}
    for rnd in range(3):
        for c in CODES:
            _lib.check(lib.accbpg_debug_gram_variant(f._h, _ptr(x), c, 20, C.byref(ms)), "variant %d" % c)
            out[c].append(ms.value)
    res = {CODES[c]: min(v) for c, v in out.items()}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
