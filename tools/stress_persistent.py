"""Repeat the persistent-kernel operator cases many times (tools/stress_persistent.py [reps]): every repetition must reproduce the first
one bit for bit and stay inside the operator tolerance - a race between LDS-DMA, staging slots and barriers would show up as a rare
mismatch rather than in the two runs of the unit tests."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_checks as C  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = C.ulib.load()
cases = [
    ("akgm8", lambda: C.akgm_case(4, 64, 96, 112, seed=3), 9),
    ("akgm16", lambda: C.akgm_case(3, 128, 64, 80, seed=4), 14),
    ("akgm32", lambda: C.akgm_case(2, 256, 48, 40, seed=5), 16),
    ("akgm32_real", lambda: C.akgm_case(5, 256, 72, 72, seed=6), 0),
    ("attn512", lambda: C.attention_case(3, 512, 18, 18, seed=7, flash=1), 0),
    ("attn256", lambda: C.attention_case(3, 256, 20, 24, seed=8, flash=1), 0),
]
# conv_sk (round 4): counted vmcnt / lgkmcnt, LDS-DMA ring, the K cut + finish pass, res_conv workgroups, Upsample classes, stream-K remainder
def _sk(kind, fn):
    def run():
        C.ulib.check(L.ucdir_debug_flag(b"convsk", kind))
        try:
            return fn()
        finally:
            C.ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    return run


cases += [
    ("sk_level4_ksplit", _sk(2, lambda: C.conv_case(16, 18, 18, 512, 512, 512, 3, 0, True, True, False, seed=5)), 0),
    ("sk_level3", _sk(2, lambda: C.conv_case(16, 36, 36, 512, 0, 512, 3, 0, True, True, False, seed=5)), 0),
    ("sk_res", _sk(2, lambda: C.conv_res_case(3, 18, 18, 512, 256, 512, seed=7)), 0),
    ("sk_up", _sk(2, lambda: C.conv_case(2, 16, 24, 128, 0, 256, 3, 2, False, False, False, seed=5)), 0),
    ("sk_streamk_7wg", _sk(2, lambda: C.conv_case(2, 36, 36, 256, 0, 512, 3, 0, True, True, True, seed=5)), 7),
    ("sk8_streamk", _sk(1, lambda: C.conv_case(5, 10, 12, 64, 0, 256, 3, 0, True, False, False, seed=5)), 3),
]
bad = 0
for name, fn, grid in cases:
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    try:
        first = None
        for r in range(reps):
            m = fn()
            key = (m["max_abs"], m["rel_rms"], str(m.get("stats")), str(m.get("stats_rel")), str(m.get("res_rel_rms")))
            if first is None:
                first = key
                assert not m["nan"] and m.get("rel_rms_branch", m["rel_rms"]) < 1.2e-2, (name, m)
            elif key != first:
                bad += 1
                print("MISMATCH", name, r, key, first)
        print(name, "ok" if bad == 0 else "BAD", first[:2])
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
print("stress done, mismatches:", bad)
sys.exit(1 if bad else 0)
