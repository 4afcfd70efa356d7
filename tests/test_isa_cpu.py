"""Guard on the generated code of the two MFMA-bound kernels (CPU only: hipcc cross-compiles without a GPU).

Both run one wavefront per SIMD with 256 accumulator registers per lane and a pinned main loop; twice in round 3 an
innocent-looking source change (a runtime `beta` in the epilogue, a loop split through a generic lambda) made the register
allocator move accumulators between the two register files INSIDE the main loop (hundreds of v_accvgpr_read / _write per
k-step, scratch spills) -- results stay right, parity tests stay green, and the kernel runs at less than half its rate.
So: the k-step loop of every production instantiation must hold its 128 MFMAs and no accumulator-file moves or scratch
traffic, and the kernels may not use more scratch than they did when they were tuned."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "accbpg_and_fw_amd", "csrc", "dopt_kernels.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# mangled-name fragments of the production kernels -> scratch bytes they are allowed (their state when tuned)
KERNELS = {
    "gram_streamk_glds_kernelINS_4TileILi256ELi128ELi64ELi128ELb0ELb0EEELi224EEE": 0,
    "gram_streamk_glds_batch_kernel": 0,
    "colnorm_glds_kernelINS_4TileILi256ELi128ELi64ELi128ELb1ELb0EEELi256EEE": 32,
    "colnorm_glds_batch_kernel": 32,
}


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "dopt_kernels.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w",
                           "-o", str(out), SRC])
    return open(out).read().split("\n")


def _kernel_lines(lines, fragment):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN6accbpg") and fragment in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    return lines[start:end]


@pytest.mark.parametrize("fragment", sorted(KERNELS))
def test_main_loop_keeps_its_accumulators(asm, fragment):
    body = _kernel_lines(asm, fragment)
    scratch = [int(l.split()[-1]) for l in body if ".amdhsa_private_segment_fixed_size" in l]
    assert scratch and scratch[0] <= KERNELS[fragment], (fragment, scratch)
    # basic blocks (label to label); the k-step loop is the one with the most MFMAs
    blocks, cur = [], []
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = []
        cur.append(l.strip().split(" ")[0].split("\t")[0] if l.strip() else "")
    blocks.append(cur)
    loop = max(blocks, key=lambda b: sum(op.startswith("v_mfma") for op in b))
    mfma = sum(op.startswith("v_mfma") for op in loop)
    moves = sum(op.startswith("v_accvgpr") for op in loop)
    spills = sum(op.startswith("scratch_") for op in loop)
    # (the gradient kernel's triangular skip splits its loop into blocks of one fragment row: 8 MFMAs)
    assert mfma == (128 if fragment.startswith("gram") else 8) or mfma >= 128, (fragment, mfma)
    assert moves == 0 and spills == 0, "%s: %d accumulator-file moves, %d scratch accesses in the k-step loop" % (fragment, moves, spills)
    total_moves = sum(l.strip().startswith("v_accvgpr_read") for l in body)
    assert total_moves <= 600, (fragment, total_moves)          # epilogues read the accumulators once


@pytest.mark.parametrize("fragment", sorted(KERNELS))
def test_compiler_m0_writes_reach_their_loads(asm, fragment):
    """The hand-written load statements (mfma_tile.hpp: glds16_run2 / _run4) write M0 themselves.  Without their "m0"
    clobber the compiler moved the M0 write of one of ITS LDS-DMA loads above such a statement, and that load landed where
    the statement had pointed M0 (wrong Gram matrices, no fault).  So: between an M0 write the compiler emitted and the
    LDS-DMA load it belongs to there may be no inline-asm statement that touches M0."""
    body = [l.strip() for l in _kernel_lines(asm, fragment)]
    in_asm, pending, runs = False, None, 0
    for i, l in enumerate(body):
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm:
            if "m0" in l:
                runs += 1
                assert pending is None, "%s: asm statement writes M0 between the compiler's '%s' and its load" % (fragment, pending)
            continue
        if re.match(r"s_\w+\s+m0,", l):
            pending = l
        elif l.startswith("global_load_lds") or l.startswith("buffer_load") and " lds" in l:
            pending = None
    assert runs > 0, "%s: no hand-written load statement found -- the check above tested nothing" % fragment
