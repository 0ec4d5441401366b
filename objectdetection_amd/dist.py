"""Scene-sharded data parallelism: one process per GPU, every rank owns
whole scenes (frames) -- each frame has its own voxels and rulebooks, so the data path needs no
collective -- and ONE exchange step per training iteration: a flat-bucket gradient all-reduce
(RCCL over xGMI with backend "nccl"; gloo on CPU for tests).  BatchNorm statistics stay per rank,
as in the reference (no SyncBN; second/pytorch/models/middle.py:124-125).

The reference's own multi-GPU path is single-process torch.nn.DataParallel
(second/pytorch/train.py:204-205); it is not reproduced.
"""
import torch
import torch.distributed as dist


def shard_scenes(costs, world_size):
    """Assign scenes to ranks so that per-rank cost (e.g. point or voxel counts) is balanced:
    longest-processing-time greedy.  -> list (per rank) of scene indices, each sorted.

    K varies per frame, so equal frame counts do not mean equal work."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], len(shards[q]), q))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


class FlatGradAllReduce:
    """All parameters' gradients live in ONE flat fp32 buffer (p.grad are views into it), so the
    exchange step is a single all-reduce of ~7.4 MB for the SECOND front end + RPN instead of one
    per tensor: over point-to-point xGMI links a ring all-reduce is latency-bound at this size and
    one large message beats many small ones."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            seg = self.flat[off:off + p.numel()]
            if p.is_contiguous():
                p.grad = seg.view_as(p)
            elif p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
                # same physical order as the parameter (autograd's gradient layout contract):
                # elementwise optimizer kernels then run on matching strides, and the fused
                # multi-tensor AdamW accepts the pair
                p.grad = seg.as_strided(p.size(), p.stride())
            else:
                raise ValueError("FlatGradAllReduce: parameters must be dense (contiguous or "
                                 "channels_last)")
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def clip_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ over all parameters, on the flat buffer: the 2-norm of
        the concatenation is the norm of the per-tensor norms.  Returns the total norm."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total

    def reduce(self, async_op=False):
        """Average gradients over ranks (sum, then divide by world size)."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        world = dist.get_world_size(self.group)
        if world == 1:
            return None
        self.flat.div_(world)
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU over gloo): gloo's device path hangs
            # on large ROCm tensors, so stage through the host.  RCCL ("nccl") reduces in place.
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=async_op)
