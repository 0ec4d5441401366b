"""CPU, world_size 2 over gloo: the scene-sharded gradient exchange (one flat all-reduce)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from objectdetection_amd.dist import FlatGradAllReduce, shard_scenes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    bucket = FlatGradAllReduce(model.parameters())
    data = torch.arange(4 * 8, dtype=torch.float32).view(4, 8) / 10
    shards = shard_scenes([3, 1, 2, 2], world)
    bucket.zero()
    x = data[shards[rank]]
    (model(x) ** 2).sum().backward()
    # per-rank loss is a SUM over its scenes; the exchange averages over ranks
    bucket.reduce()
    q.put((rank, shards, bucket.flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_equals_single_process_average():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert torch.equal(res[0][2], res[1][2])  # every rank holds the same averaged gradient
    shards = res[0][1]
    assert sorted(sum(shards, [])) == [0, 1, 2, 3]
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    data = torch.arange(4 * 8, dtype=torch.float32).view(4, 8) / 10
    (model(data) ** 2).sum().backward()
    ref = torch.cat([p.grad.flatten() for p in model.parameters()]) / world
    assert torch.allclose(res[0][2], ref, rtol=1e-5, atol=1e-6)


def _detector_worker(rank, world, port, q):
    """The real Detector's parameters (SECOND front end + RPNV2, the bucket TrainStep owns) with
    rank-dependent synthetic gradients: the sparse convs have no CPU path, so the gradients are
    written, not computed -- what is under test is the bucket, its views and the exchange."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from objectdetection_amd.train import Detector
    net = Detector("M2")
    net.rpn.to(memory_format=torch.channels_last)  # as TrainStep does: strided gradient views
    bucket = FlatGradAllReduce(net.parameters())
    bucket.zero()
    g = torch.Generator().manual_seed(100 + rank)
    for p in net.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g))
    bucket.reduce()
    named = dict(net.named_parameters())
    names = ["frontend.middle_feature_extractor.middle_conv.0.weight", "rpn.blocks.0.1.weight",
             "frontend.voxel_feature_extractor.vfe1.linear.weight"]
    q.put((rank, bucket.flat.numel(), float(bucket.flat.double().sum()),
           {n: named[n].grad.clone() for n in names},
           all(p.grad.data_ptr() >= bucket.flat.data_ptr() and
               p.grad.data_ptr() < bucket.flat.data_ptr() + bucket.flat.numel() * 4
               for p in net.parameters())))
    dist.barrier()
    dist.destroy_process_group()


def test_detector_bucket_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_detector_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 1_800_000       # ~1.85 M parameters, one flat buffer
    assert res[0][2] == res[1][2]                    # both ranks hold the same reduced bucket
    assert res[0][4] and res[1][4]                   # every p.grad is a view into the bucket
    from objectdetection_amd.train import Detector
    net = Detector("M2")
    net.rpn.to(memory_format=torch.channels_last)
    want = {}
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        for n, p in net.named_parameters():
            r = torch.randn(p.shape, generator=g)
            if n in res[0][3]:
                want[n] = want.get(n, 0) + r / world
    for n, w in want.items():
        assert torch.allclose(res[0][3][n], w, rtol=1e-6, atol=1e-7), n
        assert torch.equal(res[0][3][n], res[1][3][n])


def test_shard_scenes_balances_cost():
    costs = [120, 20, 20, 20, 20, 20, 20, 20]
    shards = shard_scenes(costs, 2)
    loads = [sum(costs[i] for i in s) for s in shards]
    assert sorted(sum(shards, [])) == list(range(8))
    assert max(loads) - min(loads) <= 20
    assert shard_scenes([1, 1, 1, 1], 4) == [[0], [1], [2], [3]]
    assert [len(s) for s in shard_scenes([5] * 64, 8)] == [8] * 8


def _segmented_worker(rank, world, port, q):
    """TrainStep's exchange (round 3): gradients produced by autograd as fresh tensors, gathered
    per segment (front | back of a small two-stage net) into the flat buffer, the back segment
    all-reduced ASYNCHRONOUSLY from a hook at the stage boundary while backward continues, the
    front segment behind backward, wait() before use -- against the blocking whole-buffer form."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for mode in ("blocking", "overlap"):
        torch.manual_seed(0)
        front = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU())
        back = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        bucket = FlatGradAllReduce(None, attach=False,
                                   segments=[list(front.parameters()), list(back.parameters())])
        data = torch.arange(4 * 8, dtype=torch.float32).view(4, 8) / 10 + rank
        fired = []
        for step in range(2):  # the second step checks that stale views are not accumulated into
            bucket.zero()
            assert all(p.grad is None for p in bucket.params)
            h = front(data)
            if mode == "overlap":
                def hook(g):
                    bucket.collect(1)
                    bucket.reduce(1, async_op=True)
                    fired.append(1)
                    return g
                h.register_hook(hook)
            (back(h) ** 2).sum().backward()
            if mode == "overlap":
                bucket.collect(0)
                bucket.reduce(0)
                bucket.wait()
            else:
                bucket.collect()
                bucket.reduce()
        assert mode == "blocking" or len(fired) == 2
        views_ok = all(p.grad.untyped_storage().data_ptr() == bucket.flat.untyped_storage().data_ptr()
                       for p in bucket.params)
        out.append((bucket.flat.clone(), views_ok))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_segment_allreduce_equals_blocking_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_segmented_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (b0, ok0), (o0, ok1) = res[0][1]
    (b1, _), (o1, _) = res[1][1]
    assert ok0 and ok1
    assert torch.equal(b0, b1) and torch.equal(o0, o1)   # ranks agree
    assert torch.equal(b0, o0)                            # overlapped == blocking, bit for bit
    # and equals the single-process average of the two ranks' gradients
    ref = []
    for rank in range(world):
        torch.manual_seed(0)
        front = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU())
        back = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        data = torch.arange(4 * 8, dtype=torch.float32).view(4, 8) / 10 + rank
        (back(front(data)) ** 2).sum().backward()
        ref.append(torch.cat([p.grad.flatten() for p in list(front.parameters()) +
                              list(back.parameters())]))
    assert torch.allclose(b0, (ref[0] + ref[1]) / world, rtol=1e-5, atol=1e-6)
