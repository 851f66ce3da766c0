"""Build-time guard for the counted-wait kernels (round-5 advice): compile csrc/engine.hip with
-Rpass-analysis=kernel-resource-usage and compare every kernel's scratch / spill figures with the
committed table profiles/kernel_resources.json.

The kernels that read LDS by inline asm with hand-counted s_waitcnt (cgemm, conv_ws*, qkv_ws, flash_attn2,
akgm_ws64, conv_sk) hold "=v" asm outputs whose data arrives later; a spill or a copy of such a register in
front of the manual wait would read a stale value without a diagnostic.  Scratch that appears in a kernel
whose table entry says 0, or that grows, fails this check.

    python tools/check_resources.py            # compare with the table (exit 1 on a regression)
    python tools/check_resources.py --update   # rewrite the table from this build (and record the compiler version)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "kernel_resources.json")


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.split("\n") if r.returncode == 0 else names
    except OSError:
        out = names
    return [re.sub(r"\(.*$", "", o) for o in out[:len(names)]]


def measure():
    sys.path.insert(0, ROOT)
    from ucdir_amd import build as b
    with tempfile.TemporaryDirectory() as d:
        cmd = [b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize",
               "-Rpass-analysis=kernel-resource-usage", os.path.join(b.CSRC, "engine.hip"), "-o", os.path.join(d, "x.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise SystemExit("hipcc failed")
    res, cur = {}, None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            key = {"ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill",
                   "Occupancy [waves/SIMD]": "occupancy"}.get(m.group(1), m.group(1).lower())
            res[cur][key] = int(m.group(2))
    names = list(res)
    pretty = demangle(names)
    ver = subprocess.run([b._hipcc(), "--version"], capture_output=True, text=True).stdout.strip().split("\n")
    return {"compiler": [v for v in ver if "version" in v.lower()][:2], "kernels": {p: res[n] for n, p in zip(names, pretty)}}


def main():
    cur = measure()
    if "--update" in sys.argv or not os.path.exists(TABLE):
        with open(TABLE, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
        print("wrote", TABLE, "(%d kernels)" % len(cur["kernels"]))
        return 0
    ref = json.load(open(TABLE))
    bad = []
    if ref.get("compiler") != cur["compiler"]:
        print("note: compiler differs from the one the table was made with:", cur["compiler"], "vs", ref.get("compiler"))
    for k, v in cur["kernels"].items():
        r = ref["kernels"].get(k)
        if r is None:
            if v.get("scratch", 0) or v.get("vgpr_spill", 0):
                bad.append("%s: new kernel with scratch %d B / %d spilled VGPRs" % (k, v.get("scratch", 0), v.get("vgpr_spill", 0)))
            continue
        if v.get("scratch", 0) > r.get("scratch", 0) or v.get("vgpr_spill", 0) > r.get("vgpr_spill", 0):
            bad.append("%s: scratch %d -> %d B per lane, spilled VGPRs %d -> %d" %
                       (k, r.get("scratch", 0), v.get("scratch", 0), r.get("vgpr_spill", 0), v.get("vgpr_spill", 0)))
    for b_ in bad:
        print("REGRESSION", b_)
    print("%d kernels checked, %d regressions" % (len(cur["kernels"]), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
