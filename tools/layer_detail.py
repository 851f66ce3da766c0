"""Format the PROF lines of `UCDIR_PROF_DETAIL=1 python bench.py --steps 1 --warmup 1` (stderr) as the per-layer table committed under profiles/:
    python tools/layer_detail.py gpurun_out/r06_detail_raw.txt > profiles/r06_layer_detail.txt
One line per (profiler key, shape): launches of that shape per forward, mean us per launch over the run's samples, TFLOP/s as bench.py counts them."""
import collections
import re
import sys

NAMES = {1: "cgemm<64,down>", 22: "conv3x3_halo<64>+res", 23: "conv_ws", 24: "conv_ws128+res", 101: "cgemm<128,down>", 105: "qkv_ws",
         113: "akgm_ws<8>", 114: "akgm_ws<16>", 115: "akgm_ws32", 116: "akgm_ws64", 120: "conv3x3_halo<128>", 121: "conv3x3_halo<128> up",
         127: "conv_sk<1,4,9>(+res)", 128: "conv_sk<1,4,4> up", 129: "conv_sk<1,4,9> wide+short(+res)", 130: "flash_attn2", 131: "flash_attn2<fp16>",
         20: "conv3x3_halo<64>", 100: "cgemm<128,s1>", 0: "cgemm<64,s1>"}
acc = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"PROF key=(\d+) H=(\d+) W=(\d+) cin=(\d+) cout=(\d+) launches=(\d+) ms=([\d.]+) TF=([\d.]+)", line)
    if not m:
        continue
    k = tuple(int(v) for v in m.groups()[:5])
    a = acc.setdefault(k, {"launches": [], "ms": [], "tf": []})
    a["launches"].append(int(m.group(6))); a["ms"].append(float(m.group(7))); a["tf"].append(float(m.group(8)))
print("# per-layer timings of a forward at B = 16, 256^2 (UCDIR_PROF_DETAIL=1 python bench.py --steps 1 --warmup 1: HIP events around every launch of the profiled")
print("# classes; one line per (profiler key, shape); 'launches' = launches of that shape per forward, 'us' = mean per launch over the run's samples; flops as bench.py counts them)")
print("# key  kernel                             H x W      cin(*taps)  cout  launches   us per launch   TFLOP/s")
for (key, H, W, cin, cout), a in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1], kv[0][3])):
    n = a["launches"][0]
    us = sum(a["ms"]) / sum(a["launches"]) * 1e3
    tf = sum(t * m for t, m in zip(a["tf"], a["ms"])) / sum(a["ms"])
    print("%4d  %-34s %4dx%-4d %9d %5d %7d %14.1f %12.1f" % (key, NAMES.get(key, str(key)), H, W, cin, cout, n, us, tf))
