import torch, time
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)
def build(n_main, n_side):
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if n_side:
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    for _ in range(n_side):
                        y.add_(1)
            for _ in range(n_main):
                x.add_(1)
            if n_side:
                s.wait_stream(side)
    return g
for n_main, n_side in [(1, 0), (22, 0), (22, 40), (62, 0), (10, 0)]:
    g = build(n_main, n_side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 500 * 1e6
    print(f"graph with {n_main} main + {n_side} side tiny kernels: {dt:.1f} us per replay")
