#!/usr/bin/env python3
"""How does the hipGraph executor (ROCm 7.2, torch.cuda.graph capture of two streams) order a
two-branch graph?  Kernels of known duration that stamp their start / end; several capture orders
of the same dependency graph; prints when each kernel ran.  tools/ubench/spin.hip must be built:
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/ubench/libspin.so tools/ubench/spin.hip
"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "ubench", "libspin.so"))
lib.spin_launch.argtypes = [ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
names = {}


def k(name, us, stream):
    slot = len(names)
    names[slot] = (name, us)
    lib.spin_launch(us, stamps.data_ptr(), slot, stream.cuda_stream)


def run(title, body):
    names.clear()
    stamps.zero_()
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        body(main, side)
    saved = dict(names)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 2)
    t0 = min(st[i, 0] for i in saved)
    print(title)
    for i in sorted(saved, key=lambda i: st[i, 0]):
        print(f"   {saved[i][0]:10s} asked {saved[i][1]:5.0f} us   ran {(st[i,0]-t0)/100:7.1f} .. "
              f"{(st[i,1]-t0)/100:7.1f}")


def fork(main, side):
    e = torch.cuda.Event()
    e.record(main)
    side.wait_event(e)


def join(main, side):
    e = torch.cuda.Event()
    e.record(side)
    main.wait_event(e)


def chain_first(main, side):
    k("root", 10, main)
    fork(main, side)
    ev = {}
    for n, us in (("s1", 10), ("s2", 50), ("s3", 10), ("s4", 50), ("s5", 10)):
        k(n, us, side)
        ev[n] = torch.cuda.Event()
        ev[n].record(side)
    k("v", 30, main)          # needs root only
    main.wait_event(ev["s1"])
    k("c1", 20, main)         # needs v, s1
    main.wait_event(ev["s3"])
    k("c2", 20, main)         # needs c1, s3
    join(main, side)


def v_first(main, side):
    k("root", 10, main)
    fork(main, side)
    k("v", 30, main)
    ev = {}
    for n, us in (("s1", 10), ("s2", 50), ("s3", 10), ("s4", 50), ("s5", 10)):
        k(n, us, side)
        ev[n] = torch.cuda.Event()
        ev[n].record(side)
    main.wait_event(ev["s1"])
    k("c1", 20, main)
    main.wait_event(ev["s3"])
    k("c2", 20, main)
    join(main, side)


def interleaved(main, side):
    k("root", 10, main)
    fork(main, side)
    k("v", 30, main)
    ev = {}

    def s(n, us):
        k(n, us, side)
        ev[n] = torch.cuda.Event()
        ev[n].record(side)
    s("s1", 10)
    main.wait_event(ev["s1"])
    k("c1", 20, main)
    s("s2", 50)
    s("s3", 10)
    main.wait_event(ev["s3"])
    k("c2", 20, main)
    s("s4", 50)
    s("s5", 10)
    join(main, side)


print("dependencies: v <- root; s1..s5 a chain <- root; c1 <- v, s1; c2 <- c1, s3")
print("ideal: v 10..40, s1 10..20, s2 20..70, s3 70..80, c1 40..60, c2 80..100, s4 80..130")
run("whole side chain enqueued first", chain_first)
run("v first, then the whole side chain", v_first)
run("side chain in pieces, each before its consumer", interleaved)
