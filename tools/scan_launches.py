"""Per-launch timing of one MIPS search (library hipEvent collector): rows, ms, TFLOP/s for every scan launch.  usage: python tools/scan_launches.py [rows] [queries]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from emdr2_amd import _native
if "--lib" in sys.argv:                                     # another build of the library (A/B)
    _i = sys.argv.index("--lib"); _native.LIB_PATH = os.path.abspath(sys.argv[_i + 1]); del sys.argv[_i:_i + 2]
if "--exp" in sys.argv:                                     # experiments build: EMDR2_MIPS_ABLATE / _KERNEL / _VARIANT switches are live
    _native.LIB_PATH = _native.LIB_PATH.replace("libemdr2_hip.so", "libemdr2_hip_exp.so")
    sys.argv.remove("--exp")
from emdr2_amd.data.emdr2_index import HipIndexShard
rows = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS_FULL
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = _native.lib()
sh = HipIndexShard(768, rows, 0)
for blk in bench.synth_rows(0, rows):
    sh.append_rows(blk)
q = torch.randn((nq, 768), device="cuda").half()
for _ in range(3):
    sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
lib.emdr2_mips_set_timing(1)
for _ in range(5):
    sh.search(q, 50, exact_fallback=False)
torch.cuda.synchronize()
cap = 256
ms = (ctypes.c_float * cap)(); rl = (ctypes.c_int64 * cap)(); n = ctypes.c_int()
_native.check(lib.emdr2_mips_timing_collect(ms, rl, cap, ctypes.byref(n)), "collect")
per = n.value // 5
for i in range(per):
    m = sum(ms[j * per + i] for j in range(5)) / 5
    print("launch %d: %9d rows  %7.3f ms  %6.0f TFLOP/s" % (i, rl[i], m, 2.0 * nq * rl[i] * 768 / (m * 1e-3) / 1e12))
print("sum of scan launches: %.3f ms" % (sum(ms[i] for i in range(n.value)) / 5))
