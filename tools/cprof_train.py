import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.getcwd())
from objectdetection_amd import synth
from objectdetection_amd.train import TrainStep
dev = torch.device("cuda:0")
step = TrainStep(dev, "M2", conv_math="bf16")
frames = [torch.from_numpy(synth.vkitti_frame(1000 + i, 20000)).to(dev) for i in range(16)]
for i in range(4):
    step(frames[:8])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(frames[8 * (i % 2):8 * (i % 2) + 8])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
