"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip (MI355X_MICROARCH.md, HBM section):
  rocprofv3 --pmc WRITE_SIZE -- python tools/calib_pmc.py     (and --pmc FETCH_SIZE in a second pass)
fill_normal_kernel WRITES exactly 1 GiB (16-byte stores per lane, nothing read); sampler_step_kernel reads 2 GiB (x_t, eps; the noise is
generated in registers) and writes 1 GiB in place.  tools/kpmc-style summaries divide the counter by these byte counts."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hip_checks as C
from ucdir_amd import lib as ulib
L = ulib.load()
n = 1 << 28                                     # fp32 elements = 1 GiB
x = torch.empty(n, device="cuda"); e = torch.zeros(n, device="cuda")
for _ in range(3):
    ulib.check(L.ucdir_fill_normal(C._p(x), n, 1234, 0, C._st()))
torch.cuda.synchronize()
print("filled", float(x[:1000].std()))
