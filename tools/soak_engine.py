import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
from objectdetection_amd import synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_
dev = torch.device("cuda:0")
model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
rng = np.random.default_rng(0)
clouds = [torch.from_numpy(synth.vkitti_frame(100 + i, int(n))).to(dev)
          for i, n in enumerate(rng.integers(2000, 24000, 24))]
with torch.no_grad():
    eng = model.engine(calib=[c for c in clouds if c.shape[0] > 20000][:3])
    t0 = time.time()
    for i in range(6000):
        c = clouds[i % len(clouds)]
        out = eng.forward(c)
        if i % 500 == 0:
            eng.check()
            ref = model([c])
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-4 and torch.isfinite(out).all(), (i, err)
            print(i, "ok err", err, "elapsed", round(time.time() - t0, 1), flush=True)
    torch.cuda.synchronize()
    eng.check()
print("soak done", round(time.time() - t0, 1))
