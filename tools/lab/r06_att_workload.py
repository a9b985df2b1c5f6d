import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sonicsim_amd import ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0); seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
out = torch.empty((sc.C, sc.T), device=dev)
for _ in range(3):
    ops.convolve_moving_seg(x, bank, seg, out=out)
torch.cuda.synchronize()
