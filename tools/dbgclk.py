import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import SonicSim_audio as A, ops, _lib
ops.init(0)
y = (0.05 * torch.randn(8, 960000, device="cuda:0")).contiguous()
for _ in range(5):
    A.get_lufs_norm_audio(y, 16000, -17, allow_many_channels=True, channel_first=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.load()._name)
buf = (ctypes.c_ulonglong * (12 * 2 * 256))()
lib.ss_debug_clk(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(12, 2, 256).astype(np.int64)
for kid, name, n in ((0, "local", 32), (1, "carry", 24), (2, "gate", 1), (3, "final", 2)):
    st, en = a[kid, 0, :n], a[kid, 1, :n]
    print(name, "span us", (en.max() - st.min()) / 100.0, "per-wg us min/max", ((en - st) / 100.0).min(), ((en - st) / 100.0).max(), "start spread", (st.max() - st.min()) / 100.0)
print("local.start -> carry.start", (a[1,0,:24].min() - a[0,0,:32].min())/100.0, "carry.end->gate.start", (a[2,0,0]-a[1,1,:24].max())/100.0)
g0=a[2,0,0]
print("gate: lbuf done %.2f | stage0 reduce %.2f thread0 %.2f | stage1 reduce %.2f thread0 %.2f | end %.2f" % tuple((x-g0)/100.0 for x in (a[4,0,0],a[5,0,0],a[6,0,0],a[5,1,0],a[6,1,0],a[2,1,0])))
n=256
s0=a[0,0,:n]; print("fused per-wg us: walk1 done %.2f sync %.2f scan done %.2f end %.2f (medians); span %.2f" % (np.median(a[1,0,:n]-s0)/100, np.median(a[1,1,:n]-s0)/100, np.median(a[7,0,:n]-s0)/100, np.median(a[0,1,:n]-s0)/100, (a[0,1,:n].max()-s0.min())/100))
# thread 64 (first wave that walks its chunk twice): entry = stamp before the barrier
e=a[8,0,:n]; print("fused, thread 64, us after its own entry (medians): barrier passed %.2f | end states %.2f | scan + start state %.2f | second walk done %.2f ; entry after wg start %.2f" % (np.median(a[8,1,:n]-e)/100, np.median(a[9,0,:n]-e)/100, np.median(a[9,1,:n]-e)/100, np.median(a[10,0,:n]-e)/100, np.median(e-s0)/100))
