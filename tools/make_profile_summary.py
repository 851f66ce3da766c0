"""Turn a rocprofv3 output directory (gpurun_out/prof_*) into the committed summaries under profiles/.

  python tools/make_profile_summary.py gpurun_out/prof_r01b r01

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim),
profiles/<tag>_bench.json (the bench line printed under the profiler), profiles/<tag>_pmc_hbm.csv
(per-kernel HBM bytes per launch from separate --pmc FETCH_SIZE / WRITE_SIZE passes) and
profiles/traffic.json (read by bench.py).  HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM:
counters are in KiB, and on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

MODE = {0: "s1", 1: "down", 2: "up", 3: "plain", 4: "s1c"}


def bench_name(k):
    m = re.search(r"cgemm_kernel<(\d+), (\d+), (\d+)>", k)
    if m:
        tm, epi, mode = int(m.group(1)), int(m.group(2)), int(m.group(3))
        return f"cgemm<{tm},akgm>" if epi == 1 else f"cgemm<{tm},std,{MODE[mode]}>"
    if "akgm_halo_kernel" in k:
        return "akgm_halo"                          # <true> / <false> instantiations share one bench row
    if "akgm_pre_kernel" in k:
        return "akgm_pre"
    m = re.search(r"conv3x3_halo_kernel<(\d+)(?:, (true|false))?>", k)
    if m:
        # <128> / <64> also carry the parity-decomposed Upsample launches; <64, true> = conv1 + fused res_conv
        return f"conv3x3_halo<{m.group(1)}>" + ("+res" if m.group(2) == "true" else "")
    return None


def agg(path, cname):
    d = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs("profiles", exist_ok=True)
    for f in os.listdir(os.path.join(src, "trace")):
        if f.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(src, "trace", f), f"profiles/{tag}_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "bench_trace.json")):
        shutil.copy(os.path.join(src, "bench_trace.json"), f"profiles/{tag}_bench.json")
    fe = agg(os.path.join(src, "pmc_fetch", [f for f in os.listdir(os.path.join(src, "pmc_fetch")) if f.endswith("counter_collection.csv")][0]), "FETCH_SIZE")
    wr = agg(os.path.join(src, "pmc_write", [f for f in os.listdir(os.path.join(src, "pmc_write")) if f.endswith("counter_collection.csv")][0]), "WRITE_SIZE")
    traffic = {}
    tsum = {}
    with open(f"profiles/{tag}_pmc_hbm.csv", "w") as out:
        out.write("kernel,launches,FETCH_SIZE_KiB_per_launch,WRITE_SIZE_KiB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k in sorted(fe, key=lambda k: -fe[k][1]):
            n = fe[k][0]
            f_ = fe[k][1] / n
            w_ = wr[k][1] / max(wr[k][0], 1) if k in wr else 0.0
            b = (2 * f_ + w_) * 1024
            out.write(f"\"{k[:90]}\",{n},{f_:.1f},{w_:.1f},{b:.0f}\n")
            bn = bench_name(k)
            if bn:                                      # several instantiations -> launch-weighted mean
                tb, tn = tsum.get(bn, (0.0, 0))
                tsum[bn] = (tb + b * n, tn + n)
                traffic[bn] = tsum[bn][0] / tsum[bn][1]
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
