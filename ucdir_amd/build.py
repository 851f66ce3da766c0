"""Build libucdir_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is built or shipped."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libucdir_hip.so")
SOURCES = ["engine.hip"]


def _deps():
    """Every file the translation unit can include: all of csrc/ plus the public header."""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".hpp", ".cpp"))]
    out.append(os.path.join(HERE, "..", "include", "ucdir_hip.h"))
    return out


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: cannot build libucdir_hip.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for p in _deps():
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, extra_flags=(), out=None, verbose=False):
    """Compile csrc/engine.hip for gfx950.  Returns the library path; ``verbose`` says whether it compiled or reused."""
    if out is None and not force and not needs_build():
        if verbose:
            print("libucdir_hip.so is newer than every source under csrc/ and include/: reused (not recompiled)")
        return LIB
    lib_out = out or LIB
    # -fno-slp-vectorize: hipcc 7.2's SLP pass packs the fp32 statistics adds of cgemm_kernel<64> into v_pk_add_f32 /
    # v_pk_fma_f32 with op_sel, and that code sums a few lanes wrongly and differently from run to run at the 288^2
    # level (found by tests/test_hip_gpu.py::test_forward_bit_reproducible_at_bench_size; per-workgroup sums dumped:
    # sum of squares bit-identical, plain sum off by 0.3 %).  Packed f32 VALU is no faster on gfx950 anyway.
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
           *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", lib_out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libucdir_hip.so")
    if verbose:
        print("compiled", lib_out, "with", " ".join(cmd[1:8]))
    return lib_out


if __name__ == "__main__":
    if "--timing" in sys.argv:        # s_memtime-stamped build for tools/ (prints per-phase cycle deltas of one workgroup)
        print(build(force=True, extra_flags=("-DUCDIR_TIMING",), out=os.path.join(HERE, "libucdir_hip_timing.so"), verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
