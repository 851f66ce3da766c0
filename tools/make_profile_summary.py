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
    if "akgm_halo_kernel" in k or "akgm_halo_stage_kernel" in k:
        return "akgm_halo"                          # <true> / <false> instantiations share one bench row
    if "akgm_pre_kernel" in k:
        return "akgm_pre"
    if "akgm_ws64_kernel" in k:
        return "akgm_ws64"
    if "akgm_ws32_kernel" in k:
        return "akgm_ws32"
    if "flash_attn2_kernel" in k or "flash_attn_kernel" in k:
        return "flash_attn<fp16>" if ", true>" in k else "flash_attn<bf16>"
    if "akgm_ws_kernel<16>" in k:
        return "akgm_ws<16>"
    if "akgm_ws_kernel" in k:
        return "akgm_ws<8>"
    if "qkv_ws_kernel" in k:
        return "qkv_ws"
    if "conv_ws128_kernel" in k:
        return "conv_ws<128->64>+res"
    if "conv_ws_kernel" in k:
        return "conv_ws<64>"
    m = re.search(r"conv_sk_kernel<(\d+), (\d+), (\d+)>", k)
    if m:                                           # <MW, waves, taps>: taps 4 = the parity classes of Upsample + conv
        waves, taps = int(m.group(2)), int(m.group(3))
        return ("upconv_sk" if taps == 4 else "conv_sk") + f"<{waves} waves>" + ("+res" if waves == 4 and taps == 9 else "")
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
    sq_summary(src, tag)


def sq_summary(src, tag):
    """Per-kernel SQ counters (separate --pmc passes) -> profiles/<tag>_pmc_sq.csv with derived fractions.
    SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per
    SIMD (= 32 x the number of 32x32x16 MFMAs, MI355X_MICROARCH.md; checked: 150.8M = 32 x 4.712M for conv3x3_halo<128>).
    GRBM_GUI_ACTIVE is summed over the 8 XCDs (2.97M per 183 us launch = 8 x 371k cycles = 2.03 GHz), so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.defaultdict(int)
    for sub in ("pmc_sq1", "pmc_sq2"):
        d = os.path.join(src, sub)
        if not os.path.isdir(d):
            continue
        for f in os.listdir(d):
            if not f.endswith("counter_collection.csv"):
                continue
            seen = collections.defaultdict(set)
            for r in csv.DictReader(open(os.path.join(d, f))):
                k = r["Kernel_Name"]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r.get("Dispatch_Id", ""))
            for k, v in seen.items():
                nl[k] = max(nl[k], len(v))
    if not tot:
        return
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
            "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU"]
    with open(f"profiles/{tag}_pmc_sq.csv", "w") as out:
        out.write("kernel,launches," + ",".join(c + "_per_launch" for c in cols) +
                  ",mfma_busy_frac,wave_wait_frac,wave_valu_frac,valu_per_mfma,lds_conflict_frac,clock_ghz_x_us\n")
        for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
            n = max(nl[k], 1)
            v = {c: tot[k].get(c, 0.0) / n for c in cols}
            gui = v["GRBM_GUI_ACTIVE"] or float("nan")
            wc = v["SQ_WAVE_CYCLES"] or float("nan")
            derived = [v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), v["SQ_WAIT_ANY"] / wc, v["SQ_ACTIVE_INST_VALU"] / wc,
                       v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"] if v["SQ_INSTS_MFMA"] else float("nan"),
                       v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"] if v["SQ_LDS_IDX_ACTIVE"] else float("nan")]
            out.write(f"\"{k[:90]}\",{n}," + ",".join("%.0f" % v[c] for c in cols) + "," + ",".join("%.4f" % x for x in derived) + "\n")
    print(open(f"profiles/{tag}_pmc_sq.csv").read()[:3000])


if __name__ == "__main__":
    main()
