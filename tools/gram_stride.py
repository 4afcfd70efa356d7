"""Does the row stride of V decide the Gram kernel's rate?  The same (m, 32768) product on column blocks of matrices with
longer rows (leading dimension 32768 ... 262144 doubles = 256 KiB ... 2 MiB between consecutive rows): a tile streams 384
rows at once, so with 2 MiB between rows every row of every workgroup sits in a page of its own."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from accbpg_and_fw_amd import _lib
    lib = _lib.load()
    m, n = 8192, 32768
    res = []
    gen = torch.Generator(device="cuda").manual_seed(2)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) + 0.05
    for ld in (32768, 65536, 131072, 262144):
        big = torch.randn(m, ld, dtype=torch.float64, device="cuda", generator=gen)
        gram = torch.empty(m, m, dtype=torch.float64, device="cuda")
        h = C.c_void_p()
        _lib.check(lib.accbpg_dopt_create(C.c_void_p(big.data_ptr()), m, n, ld, None, C.byref(h), 1), "create")
        lib.accbpg_dopt_profile_enable(h, 1)
        for rep in range(2):
            lib.accbpg_dopt_profile_reset(h)
            for _ in range(5):
                _lib.check(lib.accbpg_dopt_gram(h, C.c_void_p(x.data_ptr()), C.c_void_p(gram.data_ptr())), "gram")
            ms, cnt = C.c_double(0.0), C.c_int64(0)
            lib.accbpg_dopt_profile_read(h, 0, C.byref(ms), C.byref(cnt))
        rec = {"m": m, "n": n, "ldv": ld, "row_stride_KiB": ld * 8 // 1024, "gram_ms": ms.value / cnt.value,
               "tflops": float(m) * m * n / (ms.value / cnt.value) * 1e-9}
        print(rec, flush=True)
        res.append(rec)
        lib.accbpg_dopt_destroy(h)
        del big
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
