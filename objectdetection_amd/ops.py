"""Torch-level wrappers over the liblcf C ABI (include/lcf.h).

PyTorch is plumbing here: it owns device memory and streams; every computation is a HIP kernel
in liblcf.so enqueued on torch's current stream.  Nothing in this module has a CPU fallback:
tensors must live on the GPU, and the library must have been built.
"""
import torch
from torch.nn import functional as F

from . import _lib
from ._lib import check, f3, f6, i3

MAX_KVOL = 64


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "objectdetection_amd ops run on the GPU only (no CPU fallback); got a "
                f"{t.device} tensor")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _prod(v):
    p = 1
    for x in v:
        p *= int(x)
    return p


def to3(v):
    if isinstance(v, int):
        return [v, v, v]
    v = [int(x) for x in v]
    if len(v) != 3:
        raise ValueError(f"expected an int or 3 ints, got {v}")
    return v


# ------------------------------------------------------------------ voxelizer

def voxelize(points, voxel_size, pc_range, max_points, max_voxels, cap_mode=0, n_dev=None,
             batch_idx=None, out=None, ws=None, ctx=None):
    """points [N,F] f32 cuda -> (voxels [max_voxels,T,F], coors [max_voxels,3] zyx,
    num_points [max_voxels], voxel_num [1] i32 device).  Rows >= voxel_num are zero.
    Bit-identical to the reference's sequential loop (include/lcf.h: lcf_voxelize).

    ``n_dev`` ([1] int32 cuda): the number of valid points lives on the device (N is then the
    capacity).  ``batch_idx`` not None: coors come out as [max_voxels,4] (batch_idx,z,y,x).
    ``out`` = (voxels, coors, num, voxel_num) and ``ws`` reuse caller-owned buffers; ``ctx`` is
    an lcf_ctx handle (``_lib.Context().handle``) carrying per-call options."""
    _need_cuda(points)
    if points.dtype != torch.float32 or points.dim() != 2:
        raise ValueError("points must be a [N,F] float32 tensor")
    pts = points.contiguous()
    n, f = pts.shape
    dev = pts.device
    L = _lib.lib()
    cs = 3 if batch_idx is None else 4
    if out is None:
        voxels = torch.empty((max_voxels, max_points, f), dtype=torch.float32, device=dev)
        coors = torch.empty((max_voxels, cs), dtype=torch.int32, device=dev)
        num = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
        vnum = torch.empty((1,), dtype=torch.int32, device=dev)
    else:
        voxels, coors, num, vnum = out
    if ws is None:
        ws = _ws(L.lcf_voxelize_workspace_bytes(n, max_points), dev)
    check(L.lcf_voxelize(_ptr(pts), n, _ptr(n_dev), f, f3(voxel_size), f6(pc_range),
                         int(max_points), int(max_voxels), int(cap_mode), int(batch_idx or 0), cs,
                         _ptr(voxels), _ptr(coors), _ptr(num), _ptr(vnum), _ptr(ws), ws.numel(),
                         _stream(), ctx))
    return voxels, coors, num, vnum


class VoxelAccumulator:
    """A cumulative point cloud and its voxelisation, both resident on the device (SURVEY.md 8f-2;
    include/lcf.h: lcf_voxelize_append).  The light-curtain loop stacks each curtain's returns
    under the cloud and voxelises everything again (second/light_curtain/utils.py:121-124);
    ``append`` voxelises only the new points and leaves ``voxels / coors / num_points /
    voxel_num`` equal to a from-scratch ``voxelize`` of the whole cloud.

        acc = VoxelAccumulator(capacity, 4, voxel_size, pc_range, max_points, max_voxels)
        acc.reset(); acc.append(lidar_points); ...; acc.append(curtain_points)
        voxels, coors, num, vnum = acc.outputs            # device tensors, padded to max_voxels
    """

    def __init__(self, capacity, num_features, voxel_size, pc_range, max_points, max_voxels,
                 cap_mode=0, batch_idx=None, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("objectdetection_amd ops need a GPU: no CPU path")
        L = _lib.lib()
        self.capacity, self.f = int(capacity), int(num_features)
        self.voxel_size, self.pc_range = f3(voxel_size), f6(pc_range)
        self.max_points, self.max_voxels = int(max_points), int(max_voxels)
        self.cap_mode, self.batch_idx = int(cap_mode), batch_idx
        self.cs = 3 if batch_idx is None else 4
        self.points = torch.empty((self.capacity, self.f), dtype=torch.float32, device=dev)
        self.voxels = torch.empty((self.max_voxels, self.max_points, self.f), dtype=torch.float32,
                                  device=dev)
        self.coors = torch.empty((self.max_voxels, self.cs), dtype=torch.int32, device=dev)
        self.num = torch.empty((self.max_voxels,), dtype=torch.int32, device=dev)
        self.vnum = torch.empty((1,), dtype=torch.int32, device=dev)
        self.ws = _ws(L.lcf_voxelize_workspace_bytes(self.capacity, self.max_points), dev)
        self.n = 0

    def reset(self):
        """Forget the cloud; the next append starts a new session."""
        self.n = 0

    @property
    def outputs(self):
        return self.voxels, self.coors, self.num, self.vnum

    @property
    def cloud(self):
        return self.points[:self.n]

    def append(self, new_points, ctx=None):
        """new_points [M, F] float32 (device tensor, or host array: one H2D copy of the new rows
        only).  Returns ``outputs``."""
        if not torch.is_tensor(new_points):
            new_points = torch.from_numpy(new_points)
        if new_points.dim() != 2 or new_points.shape[1] != self.f or \
                new_points.dtype != torch.float32:
            raise ValueError("new_points must be [M,%d] float32" % self.f)
        m = new_points.shape[0]
        if self.n + m > self.capacity:
            raise ValueError("cloud of %d points exceeds the accumulator's capacity %d"
                             % (self.n + m, self.capacity))
        self.points[self.n:self.n + m].copy_(new_points, non_blocking=True)
        check(_lib.lib().lcf_voxelize_append(
            _ptr(self.points), self.capacity, self.n, self.n + m, self.f, self.voxel_size,
            self.pc_range, self.max_points, self.max_voxels, self.cap_mode,
            int(self.batch_idx or 0), self.cs, _ptr(self.voxels), _ptr(self.coors),
            _ptr(self.num), _ptr(self.vnum), _ptr(self.ws), self.ws.numel(), _stream(), ctx))
        self.n += m
        return self.outputs


# ------------------------------------------------------------------ rulebooks

def conv_out_shape(shape, ksize, stride, pad, dil):
    out = i3(0)
    _lib.lib().lcf_conv_out_shape(i3(shape), i3(ksize), i3(stride), i3(pad), i3(dil), out)
    return [int(x) for x in out]


class Rulebook:
    """Native rulebook of one sparse convolution: neighbour tables (see include/lcf.h) plus,
    lazily, spconv's canonical ``indice_pairs`` / ``indice_pair_num``.

    Tuple-compatible with what spconv stores in ``SparseConvTensor.indice_dict``:
    ``(outids, indices, indice_pairs, indice_pair_num, spatial_shape)``.
    """

    def __init__(self, subm, in_indices, out_indices, in_shape, out_shape, ksize, stride,
                 padding, dilation, out_nbr, in_nbr):
        self.subm = subm
        self.in_indices = in_indices
        self.out_indices = out_indices
        self.in_shape = list(in_shape)
        self.out_shape = list(out_shape)
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        self.kvol = _prod(ksize)
        self.out_nbr = out_nbr
        self._in_nbr = in_nbr
        self._pairs = None

    @property
    def k_in(self):
        return self.in_indices.shape[0]

    @property
    def k_out(self):
        return self.out_indices.shape[0]

    @property
    def in_nbr(self):
        if self._in_nbr is None:  # submanifold: the mirror image of out_nbr
            t = torch.empty_like(self.out_nbr)
            check(_lib.lib().lcf_rulebook_mirror(_ptr(self.out_nbr), self.k_in, self.kvol,
                                                 _ptr(t), _stream()))
            self._in_nbr = t
        return self._in_nbr

    def _build_pairs(self):
        if self._pairs is None:
            L = _lib.lib()
            dev = self.in_indices.device
            pairs = torch.empty((2, self.kvol, self.k_in), dtype=torch.int32, device=dev)
            pnum = torch.empty((self.kvol,), dtype=torch.int32, device=dev)
            ws = _ws(L.lcf_rulebook_pairs_workspace_bytes(self.k_in, self.kvol), dev)
            check(L.lcf_rulebook_pairs(_ptr(self.in_nbr), self.k_in, self.kvol, _ptr(pairs),
                                       _ptr(pnum), _ptr(ws), ws.numel(), _stream()))
            self._pairs = (pairs, pnum)
        return self._pairs

    @property
    def indice_pairs(self):
        return self._build_pairs()[0]

    @property
    def indice_pair_num(self):
        return self._build_pairs()[1]

    def as_tuple(self):
        return (self.out_indices, self.in_indices, self.indice_pairs, self.indice_pair_num,
                self.out_shape)

    def __iter__(self):
        return iter(self.as_tuple())

    def __getitem__(self, i):
        return self.as_tuple()[i]

    def __len__(self):
        return 5


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm):
    """indices [K,4] i32 cuda (b,z,y,x) -> Rulebook.  spconv.ops.get_indice_pairs equivalent."""
    _need_cuda(indices)
    if indices.dtype != torch.int32 or indices.dim() != 2 or indices.shape[1] != 4:
        raise ValueError("indices must be [K,4] int32 (batch, z, y, x)")
    ind = indices.contiguous()
    k = ind.shape[0]
    dev = ind.device
    L = _lib.lib()
    shape = to3(list(spatial_shape))
    ksize, stride, padding, dilation = to3(ksize), to3(stride), to3(padding), to3(dilation)
    kvol = _prod(ksize)
    if kvol > MAX_KVOL:
        raise RuntimeError(f"kernel volume {kvol} > {MAX_KVOL} is not supported")
    if subm:
        nbr = torch.empty((k, kvol), dtype=torch.int32, device=dev)
        ws = _ws(L.lcf_rulebook_subm_workspace_bytes(k), dev)
        check(L.lcf_rulebook_subm(_ptr(ind), k, None, int(batch_size), i3(shape), i3(ksize),
                                  i3(dilation), _ptr(nbr), _ptr(ws), ws.numel(), _stream()))
        return Rulebook(True, ind, ind, shape, shape, ksize, [1, 1, 1],
                        [(a // 2) * d for a, d in zip(ksize, dilation)], dilation, nbr, None)
    oshape = conv_out_shape(shape, ksize, stride, padding, dilation)
    if min(oshape) <= 0:
        raise ValueError(f"sparse conv output shape {oshape} is empty")
    cap = L.lcf_rulebook_conv_max_out(k, i3(ksize), i3(stride))
    ws = _ws(L.lcf_rulebook_conv_workspace_bytes(k, i3(ksize), i3(stride)), dev)
    out_c = torch.empty((max(cap, 1), 4), dtype=torch.int32, device=dev)
    kout_t = torch.empty((1,), dtype=torch.int32, device=dev)
    args = (_ptr(ind), k, None, int(batch_size), i3(shape), i3(ksize), i3(stride), i3(padding),
            i3(dilation))
    check(L.lcf_rulebook_conv_discover(*args, _ptr(out_c), int(max(cap, 1)), _ptr(kout_t), None,
                                       _ptr(ws), ws.numel(), _stream()))
    k_out = int(kout_t.item())  # the one host sync per strided conv, as in spconv
    out_nbr = torch.empty((k_out, kvol), dtype=torch.int32, device=dev)
    in_nbr = torch.empty((k, kvol), dtype=torch.int32, device=dev)
    check(L.lcf_rulebook_conv_fill(*args, k_out, None, _ptr(out_nbr), _ptr(in_nbr), _ptr(ws),
                                   ws.numel(), _stream()))
    return Rulebook(False, ind, out_c[:k_out].clone(), shape, oshape, ksize, stride, padding,
                    dilation, out_nbr, in_nbr)


class RulebookChain:
    """All rulebooks of a SparseSequential in one call (include/lcf.h: lcf_rulebook_chain), with
    caller-owned buffers sized once for capacities -- what FrameEngine replays per frame.

    specs: the convs in network order, dicts with subm, ksize, stride, padding, dilation.
    caps: output-row capacity of every strided conv (None = its worst case).  After ``run`` each
    level i holds ``out_nbr[i]`` [rows_cap, kvol]; strided levels also ``out_coors[i]``,
    ``k_out[i]`` (device count) and their entry of ``flags`` (overflow)."""

    def __init__(self, specs, k_cap, batch_size, spatial_shape, caps=None, device=None,
                 plan_rows=None):
        """plan_rows: {level: tile height} -- levels whose conv tile plan
        (lcf_conv_tile_plan) the chain should build as well (``plan[level]``)."""
        L = _lib.lib()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.L, self.dev = L, dev
        self.k_cap, self.batch, self.shape = int(k_cap), int(batch_size), to3(list(spatial_shape))
        self.n = len(specs)
        self.levels = (_lib.RbLevel * self.n)()
        self.out_nbr, self.out_coors, self.k_out, self.ws = {}, {}, {}, {}
        self.plan, self.plan_rows = {}, {}
        self.rows_cap, self.kvol, self.out_shape = [], [], []
        self.strided = [i for i, sp in enumerate(specs) if not sp["subm"]]
        self.flags = torch.zeros(max(len(self.strided), 1), dtype=torch.int32, device=dev)
        cap, shape, si = self.k_cap, list(self.shape), 0
        i32 = torch.int32
        for i, sp in enumerate(specs):
            lv = self.levels[i]
            ks, dl = to3(sp["ksize"]), to3(sp.get("dilation", 1))
            st, pd = to3(sp.get("stride", 1)), to3(sp.get("padding", 0))
            lv.subm = int(bool(sp["subm"]))
            lv.ksize, lv.stride, lv.pad, lv.dil = i3(ks), i3(st), i3(pd), i3(dl)
            kvol = _prod(ks)
            self.kvol.append(kvol)
            if sp["subm"]:
                self.out_nbr[i] = torch.empty((max(cap, 1), kvol), dtype=i32, device=dev)
            else:
                worst = L.lcf_rulebook_conv_max_out(cap, i3(ks), i3(st))
                shape = conv_out_shape(shape, ks, st, pd, dl)
                cells = self.batch * shape[0] * shape[1] * shape[2]
                co = min(worst, cells) if caps is None else min(int(caps[si]), worst, cells)
                co = max(co, 64)
                nbytes = L.lcf_rulebook_chain_workspace_bytes(cap, i3(ks), i3(st), co)
                if nbytes == 0:
                    raise RuntimeError("rulebook chain: level too large for 32-bit stamps")
                self.ws[i] = _ws(nbytes, dev)
                self.out_coors[i] = torch.empty((co, 4), dtype=i32, device=dev)
                self.k_out[i] = torch.zeros(1, dtype=i32, device=dev)
                self.out_nbr[i] = torch.empty((co, kvol), dtype=i32, device=dev)
                lv.out_coors = self.out_coors[i].data_ptr()
                lv.cap_out = co
                lv.k_out = self.k_out[i].data_ptr()
                lv.overflow = self.flags[si:si + 1].data_ptr()
                lv.ws, lv.ws_bytes = self.ws[i].data_ptr(), self.ws[i].numel()
                si += 1
                cap = co
            lv.out_nbr = self.out_nbr[i].data_ptr()
            self.rows_cap.append(cap)
            self.out_shape.append(list(shape))
            if plan_rows and i in plan_rows:
                tr = int(plan_rows[i])
                self.plan[i] = torch.zeros(L.lcf_conv_tile_plan_ints(cap, kvol, tr), dtype=i32,
                                           device=dev)
                self.plan_rows[i] = tr
                lv.plan, lv.plan_tile_rows = self.plan[i].data_ptr(), tr
        self.ws0 = _ws(L.lcf_rulebook_subm_workspace_bytes(self.k_cap), dev)
        # the part of the walk (lcf_rulebook_chain_parts) that completes each level's table
        self.num_parts = L.lcf_rulebook_chain_num_parts(self.levels, self.n)
        self.part_of_level, part = [], 0
        for sp in specs:
            part += 0 if sp["subm"] else 1
            self.part_of_level.append(part)

    def set_event(self, i, event):
        """Record ``event`` (a torch.cuda.Event that has been recorded once, so that its handle
        exists) when level i's table is complete."""
        self.levels[i].ready_event = event.cuda_event

    def reset(self, stream=None, ctx=None):
        check(self.L.lcf_rulebook_chain_reset(self.levels, self.n, self.k_cap, _ptr(self.ws0),
                                              self.ws0.numel(),
                                              _stream() if stream is None else stream, ctx))

    def run(self, indices, k_dev=None, stream=None, ctx=None, parts=None):
        """indices [k_cap (or fewer, without k_dev), 4] i32 cuda.  parts = (first, last): only
        those pieces of the walk (``part_of_level``), for interleaving with the consumers."""
        k = self.k_cap if k_dev is not None else int(indices.shape[0])
        if k > self.k_cap:
            raise ValueError("rulebook chain: more rows than the planned capacity")
        p0, p1 = (0, 0x7FFFFFFF) if parts is None else parts
        check(self.L.lcf_rulebook_chain_parts(_ptr(indices), k, _ptr(k_dev), self.batch,
                                              i3(self.shape), self.levels, self.n,
                                              _ptr(self.ws0), self.ws0.numel(), int(p0), int(p1),
                                              _stream() if stream is None else stream, ctx))


# ------------------------------------------------------------------ indice_conv

def weight_image(weight, transpose=False):
    """weight [kD,kH,kW,Cin,Cout] (reference layout) -> MFMA weight image (see conv.hip)."""
    _need_cuda(weight)
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    w = w.contiguous()
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    kvol = w.numel() // (cin * cout)
    L = _lib.lib()
    n = L.lcf_weight_image_elems(kvol, cin, cout, int(transpose))
    if n == 0:
        raise RuntimeError(f"unsupported channel counts for indice_conv: {cin}->{cout}")
    img = torch.empty((n,), dtype=torch.float32, device=w.device)
    check(L.lcf_weight_image(_ptr(w), kvol, cin, cout, int(transpose), _ptr(img), _stream()))
    return img


_CONV_MATH = ["f32"]


class conv_math:
    """Context manager / setter for the MFMA operand type of indice_conv: "f32" (default, fp32
    MFMA); "bf16" for TRAINING (forward + dgrad through ``indice_conv`` with bf16 operands, fp32
    accumulate, tolerance 2e-2); "bf16x3" for INFERENCE (the no-grad forward of the sparse conv
    modules with split hi + lo bf16 operands, fp32 tolerance; "bf16x8": three exact planes and
    eight products, fp32-grade; training under either stays fp32).
    Layers whose channels are not 32/64/128 stay fp32 in every mode."""

    def __init__(self, mode):
        if mode not in ("f32", "bf16", "bf16x3", "bf16x8"):
            raise ValueError("conv math mode must be 'f32', 'bf16', 'bf16x3' or 'bf16x8'")
        self.mode, self.prev = mode, None

    def __enter__(self):
        self.prev = _CONV_MATH[0]
        _CONV_MATH[0] = self.mode
        return self

    def __exit__(self, *exc):
        _CONV_MATH[0] = self.prev


def get_conv_math():
    return _CONV_MATH[0]


def set_conv_math(mode):
    conv_math(mode).__enter__()


def bf16_supported(cin, cout):
    return cin in (32, 64, 128) and cout in (32, 64, 128)


def weight_image_bf16(weight, transpose=False):
    """weight [...,Cin,Cout] -> bf16 MFMA weight image (int16 storage) for lcf_indice_conv_fwd_bf16."""
    _need_cuda(weight)
    w = weight.detach().float().contiguous()
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    kvol = w.numel() // (cin * cout)
    img = torch.empty((kvol * cin * cout,), dtype=torch.int16, device=w.device)
    check(_lib.lib().lcf_weight_image_bf16(_ptr(w), kvol, cin, cout, int(transpose), _ptr(img),
                                           _stream()))
    return img


def conv_forward_bf16(features, wimg, nbr, n_rows, kvol, cin, cout, scale=None, shift=None,
                      relu=False, n_rows_dev=None, out=None):
    """conv_forward with bf16 MFMA operands (fp32 accumulate, fp32 in/out)."""
    _need_cuda(features, wimg, nbr)
    x = features.contiguous()
    if out is None:
        out = torch.empty((n_rows, cout), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_fwd_bf16(_ptr(x), _ptr(wimg), _ptr(nbr), int(n_rows),
                                              _ptr(n_rows_dev), int(kvol), int(cin), int(cout),
                                              _ptr(scale), _ptr(shift), int(bool(relu)),
                                              _ptr(out), _stream()))
    return out


def weight_image_bf16x3(weight, transpose=False, planes=2):
    """weight [...,Cin,Cout] -> two-plane (hi, lo) bf16 image for lcf_indice_conv_fwd_bf16x3, or
    (planes = 3) the three-plane image for lcf_indice_conv_fwd_bf16x8."""
    _need_cuda(weight)
    w = weight.detach().float().contiguous()
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    kvol = w.numel() // (cin * cout)
    img = torch.empty((planes * kvol * cin * cout,), dtype=torch.int16, device=w.device)
    L = _lib.lib()
    fn = L.lcf_weight_image_bf16x8 if planes == 3 else L.lcf_weight_image_bf16x3
    check(fn(_ptr(w), kvol, cin, cout, int(transpose), _ptr(img), _stream()))
    return img


def conv_forward_bf16x3(features, wimg, nbr, n_rows, kvol, cin, cout, scale=None, shift=None,
                        relu=False, n_rows_dev=None, out=None, dense_coors=None, batch_size=1,
                        shape=None, planes=2):
    """conv_forward at fp32 tolerance on the bf16 matrix cores: operands split into hi + lo bf16
    planes, three products per term, fp32 accumulation (include/lcf.h).  With ``dense_coors`` the
    rows land in the dense [batch, cout, D, H, W] tensor (``out`` must then be cleared)."""
    _need_cuda(features, wimg, nbr)
    x = features.contiguous()
    if dense_coors is not None:
        if out is None:
            out = torch.zeros((int(batch_size), cout, *shape), dtype=torch.float32,
                              device=x.device)
        dc, sh = dense_coors.contiguous(), i3(shape)
    else:
        if out is None:
            out = torch.empty((n_rows, cout), dtype=torch.float32, device=x.device)
        dc, sh = None, None
    L = _lib.lib()
    fn = L.lcf_indice_conv_fwd_bf16x8 if planes == 3 else L.lcf_indice_conv_fwd_bf16x3
    check(fn(
        _ptr(x), _ptr(wimg), _ptr(nbr), int(n_rows), _ptr(n_rows_dev), int(kvol), int(cin),
        int(cout), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(out), _ptr(dc),
        int(batch_size), sh, _stream()))
    return out


def conv_forward(features, wimg, nbr, n_rows, kvol, cin, cout, scale=None, shift=None,
                 relu=False, n_rows_dev=None, out=None):
    """out[n_rows,cout] = act((sum_o features[nbr[n][o]] @ W[o]) * scale + shift).

    ``nbr=None`` with ``kvol=1`` is the identity table (a plain row-wise linear layer).
    ``n_rows_dev`` ([1] int32 cuda): the actual row count lives on the device, ``n_rows`` is
    then the capacity (rows past the device count are left untouched)."""
    _need_cuda(features, wimg, nbr)
    x = features.contiguous()
    if out is None:
        out = torch.empty((n_rows, cout), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_fwd(_ptr(x), _ptr(wimg), _ptr(nbr), int(n_rows),
                                         _ptr(n_rows_dev), int(kvol), int(cin), int(cout),
                                         _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(out),
                                         _stream()))
    return out


def tile_plan(nbr, n_rows, kvol, k_dev=None, shape=None):
    """nbr [rows_cap, kvol] -> (plan, tile_rows): the compacted kernels' per-tile set-up for this
    table (include/lcf.h: lcf_conv_tile_plan), shared by every conv over it.  ``shape`` =
    (cin, cout, math) of the convs that will read it picks their tile height."""
    _need_cuda(nbr)
    L = _lib.lib()
    if shape is not None:
        tr = L.lcf_conv_tile_rows_for(int(n_rows), int(kvol), int(shape[0]), int(shape[1]),
                                      int(shape[2]))
    else:
        tr = L.lcf_conv_tile_rows(int(n_rows))
    plan = torch.zeros(L.lcf_conv_tile_plan_ints(int(n_rows), int(kvol), tr), dtype=torch.int32,
                       device=nbr.device)
    check(L.lcf_conv_tile_plan(_ptr(nbr), int(n_rows), _ptr(k_dev), int(kvol), tr, _ptr(plan),
                               _stream()))
    return plan, tr


def conv_forward_planned(features, wimg, plan, tile_rows, n_rows, kvol, cin, cout, scale=None,
                         shift=None, relu=False, math=0, out=None, k_dev=None, ctx=None):
    """lcf_indice_conv_fwd_planned: the forward over a tile plan; math 0 fp32, 1 bf16, 2 bf16x3,
    3 bf16x8 (wimg in that form).  Bit-identical to the table forward of the same form."""
    _need_cuda(features, wimg, plan)
    x = features.contiguous()
    if out is None:
        out = torch.empty((int(n_rows), cout), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_fwd_planned(
        _ptr(x), _ptr(wimg), int(math), _ptr(plan), int(tile_rows), int(n_rows), _ptr(k_dev),
        int(kvol), int(cin), int(cout), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(out), None,
        1, i3(0), _stream(), ctx))
    return out


def conv_forward_dense(features, wimg, nbr, n_rows, kvol, cin, cout, out_coors, batch_size,
                       shape, scale=None, shift=None, relu=False, n_rows_dev=None, out=None):
    """conv_forward whose rows land in a dense [batch, cout, D, H, W] tensor at ``out_coors``
    ([n_rows,4] int32 b,z,y,x): conv + SparseConvTensor.dense() in one kernel
    (lcf_indice_conv_fwd_dense).  ``out``, when given, must already be cleared."""
    _need_cuda(features, wimg, nbr, out_coors)
    x = features.contiguous()
    if out is None:
        out = torch.zeros((int(batch_size), cout, *shape), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_fwd_dense(
        _ptr(x), _ptr(wimg), _ptr(nbr), int(n_rows), _ptr(n_rows_dev), int(kvol), int(cin),
        int(cout), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(out_coors.contiguous()),
        int(batch_size), i3(shape), _ptr(out), _stream()))
    return out


def wgrad_mc_supported(cin, cout):
    return bf16_supported(cin, cout) and cin * cout <= 4096


def conv_wgrad(features, dout, in_nbr, kvol, cin, cout, planes=0):
    """dw [kvol,cin,cout].  planes 0: fp32 MFMA; 1: bf16 operands; 2: split hi + lo bf16 operands
    (fp32-grade) -- 1 and 2 on the bf16 matrix cores (lcf_indice_conv_wgrad_bf16)."""
    x = features.contiguous()
    g = dout.contiguous()
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=x.device)
    if planes:
        check(_lib.lib().lcf_indice_conv_wgrad_bf16(_ptr(x), _ptr(g), _ptr(in_nbr), x.shape[0],
                                                    int(kvol), int(cin), int(cout), int(planes),
                                                    _ptr(dw), _stream()))
        return dw
    check(_lib.lib().lcf_indice_conv_wgrad(_ptr(x), _ptr(g), _ptr(in_nbr), x.shape[0],
                                           int(kvol), int(cin), int(cout), _ptr(dw), _stream()))
    return dw


def linear_wgrad(x, dy):
    """x [n, cin], dy [n, cout] f32 cuda -> x^T dy [cin, cout] (lcf_linear_wgrad)."""
    _need_cuda(x, dy)
    x, dy = x.contiguous(), dy.contiguous()
    n, cin = x.shape
    cout = dy.shape[1]
    dw = torch.empty((cin, cout), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_linear_wgrad(_ptr(x), _ptr(dy), int(n), int(cin), int(cout), _ptr(dw),
                                      _stream()))
    return dw


def linear_wgrad_supported(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.dim() == 2 and max(weight.shape) <= 128)


class TallLinearFunction(torch.autograd.Function):
    """y = x W^T (+ b) for a tall x ([n, in] with n ~ 10^5 rows, in / out <= 128): forward and
    input gradient are library GEMMs, the weight gradient -- a reduction over all n rows into an
    [out, in] matrix, which library GEMMs serve with a handful of workgroups (0.18-0.36 ms per
    VFE layer at 160k rows) -- runs on liblcf's row-chunked MFMA wgrad kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = linear_wgrad(dy, x) if ctx.needs_input_grad[1] else None  # [out, in]
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def tall_linear(lin, x):
    """nn.Linear ``lin`` applied to x [n, in]; TallLinearFunction where it applies."""
    if torch.is_grad_enabled() and lin.weight.requires_grad and linear_wgrad_supported(x, lin.weight):
        return TallLinearFunction.apply(x, lin.weight, lin.bias)
    return lin(x)


class IndiceConvFunction(torch.autograd.Function):
    """indice_conv forward / dgrad / wgrad (torch.ops.spconv.indice_conv{,_backward})."""

    @staticmethod
    def forward(ctx, features, weight, rulebook):
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        x = features.float() if features.dtype != torch.float32 else features
        ctx.bf16 = _CONV_MATH[0] == "bf16" and bf16_supported(cin, cout)
        if ctx.bf16:
            out = conv_forward_bf16(x, weight_image_bf16(weight), rulebook.out_nbr,
                                    rulebook.k_out, rulebook.kvol, cin, cout)
        else:
            out = conv_forward(x, weight_image(weight), rulebook.out_nbr, rulebook.k_out,
                               rulebook.kvol, cin, cout)
        ctx.save_for_backward(x, weight)
        ctx.rulebook = rulebook
        ctx.in_dtype = features.dtype
        return out.to(features.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        rb = ctx.rulebook
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        g = grad_out.float().contiguous()
        din = dw = None
        if ctx.needs_input_grad[0] and ctx.bf16:
            din = conv_forward_bf16(g, weight_image_bf16(weight, transpose=True), rb.in_nbr,
                                    rb.k_in, rb.kvol, cout, cin).to(ctx.in_dtype)
        elif ctx.needs_input_grad[0]:
            din = conv_forward(g, weight_image(weight, transpose=True), rb.in_nbr, rb.k_in,
                               rb.kvol, cout, cin).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            # "bf16" conv math: the wide submanifold layers (many pairs per 64 rows) take the
            # matrix-core wgrad with split hi + lo operands -- fp32-grade gradients, 400 -> 300 us
            # per 64->64 call at batch 8; strided and narrower layers are faster on fp32 MFMAs
            planes = 2 if (ctx.bf16 and rb.subm and cin * cout >= 4096
                           and wgrad_mc_supported(cin, cout)) else 0
            dw = conv_wgrad(x, g, rb.in_nbr, rb.kvol, cin, cout,
                            planes).view_as(weight).to(weight.dtype)
        return din, dw, None


def indice_conv(features, weight, rulebook):
    _need_cuda(features, weight)
    return IndiceConvFunction.apply(features, weight, rulebook)


# ------------------------------------------------------------------ bf16-row training convs

def conv_forward_b16(features, wimg, nbr, n_rows, kvol, cin, cout, out_bf16=False):
    """indice_conv over rows STORED as bf16 (include/lcf.h: lcf_indice_conv_fwd_b16): features
    [k_in, cin] bfloat16, wimg = weight_image_bf16 -> [n_rows, cout] float32, or bfloat16 with
    ``out_bf16`` (dgrad: pass in_nbr, the transposed image and cin / cout exchanged)."""
    _need_cuda(features, wimg, nbr)
    if features.dtype != torch.bfloat16:
        raise ValueError("conv_forward_b16 takes bfloat16 rows")
    x = features.contiguous()
    out = torch.empty((int(n_rows), int(cout)),
                      dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_fwd_b16(_ptr(x), x.shape[0], _ptr(wimg), _ptr(nbr), int(n_rows),
                                             int(kvol), int(cin), int(cout), _ptr(out),
                                             int(bool(out_bf16)), _stream()))
    return out


def conv_wgrad_b16(features, dout, in_nbr, kvol, cin, cout):
    """dw [kvol, cin, cout] float32 from bfloat16 rows (lcf_indice_conv_wgrad_b16)."""
    _need_cuda(features, dout, in_nbr)
    if features.dtype != torch.bfloat16 or dout.dtype != torch.bfloat16:
        raise ValueError("conv_wgrad_b16 takes bfloat16 rows")
    x, g = features.contiguous(), dout.contiguous()
    dw = torch.empty((int(kvol), int(cin), int(cout)), dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_indice_conv_wgrad_b16(_ptr(x), _ptr(g), _ptr(in_nbr), x.shape[0],
                                               int(kvol), int(cin), int(cout), _ptr(dw),
                                               _stream()))
    return dw


def x6_conv_supported(cin, cout, kvol=27):
    """Layers lcf_indice_conv_fwd_x6 is built for."""
    return (cin, cout) in ((32, 32), (32, 64), (64, 64)) and kvol > 1


def split_planes(features, n_rows_dev=None, plane_rows=None, out=None):
    """fp32 rows [k, c] -> three exact bf16 planes [3, plane_rows, c] (lcf_split_planes): an fp32
    value is h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)."""
    _need_cuda(features)
    x = features.contiguous()
    k, c = int(x.shape[0]), int(x.shape[1])
    rows = k if plane_rows is None else int(plane_rows)
    if out is None:
        out = torch.zeros((3, rows, c), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().lcf_split_planes(_ptr(x), k, _ptr(n_rows_dev), c, _ptr(out), rows, _stream()))
    return out


def conv_forward_x6(planes, wimg3, nbr, n_rows, kvol, cin, cout, scale=None, shift=None,
                    relu=False, n_rows_dev=None, out_planes=False, out=None):
    """indice_conv on three exact bf16 planes with six products (include/lcf.h:
    lcf_indice_conv_fwd_x6): planes [3, rows, cin] bfloat16 (split_planes or a previous call with
    ``out_planes``), wimg3 = weight_image_bf16x3(weight, planes=3) -> [n_rows, cout] float32, or
    [3, n_rows, cout] bfloat16 planes for the next such conv.  fp32-grade results."""
    _need_cuda(planes, wimg3, nbr)
    if planes.dtype != torch.bfloat16 or planes.dim() != 3 or planes.shape[0] != 3:
        raise ValueError("conv_forward_x6 takes [3, rows, cin] bfloat16 planes")
    x = planes.contiguous()
    if out is None:
        out = (torch.zeros((3, int(n_rows), int(cout)), dtype=torch.bfloat16, device=x.device)
               if out_planes else
               torch.empty((int(n_rows), int(cout)), dtype=torch.float32, device=x.device))
    out_rows = int(out.shape[1]) if out_planes else int(n_rows)
    check(_lib.lib().lcf_indice_conv_fwd_x6(_ptr(x), int(x.shape[1]), _ptr(wimg3), _ptr(nbr),
                                            int(n_rows), _ptr(n_rows_dev), int(kvol), int(cin),
                                            int(cout), _ptr(scale), _ptr(shift), int(bool(relu)),
                                            _ptr(out), int(bool(out_planes)), out_rows, _stream()))
    return out


def b16_conv_supported(cin, cout):
    """Channel pairs the bf16-row training kernels (forward, dgrad and wgrad) are built for."""
    return cin in (32, 64) and cout in (32, 64)


class ConvBNReLUTrainFunction(torch.autograd.Function):
    """One ``sparse conv -> BatchNorm1d (training) -> ReLU`` triple of the middle extractor as ONE
    autograd node on bf16 rows (BASELINE.json configs[2]; second/pytorch/models/middle.py:145-192
    in training mode, second/pytorch/train.py:316-343).

    The tensors a sparse conv GATHERS -- its input y_prev and, in backward, the gradient of its own
    output -- are bf16 rows (128 bytes at 64 channels), written once by the BatchNorm kernels;
    the conv output x stays fp32 (BatchNorm statistics and its backward need it) and never leaves
    this node, so autograd's dtype rules do not force fp32 onto the gathered gradients.
        forward : x = conv(y_prev) [fp32] ; y = relu(bn(x)) [bf16 or fp32]
        backward: dx = bn_relu_bwd(x, y, dy) [bf16] ; dy_prev = dgrad(dx) [bf16] ; dW = wgrad
    Same arithmetic as IndiceConvFunction + BNReLUFunction under conv_math("bf16") in forward and
    dgrad (those round the same operands to bf16 when they enter LDS); the weight gradient takes
    its operands at bf16 as well (the split hi + lo form needs fp32 rows)."""

    @staticmethod
    def forward(ctx, feats, weight, gamma, beta, rulebook, eps, relu, run, y_bf16):
        L = _lib.lib()
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        rb = rulebook
        x = conv_forward_b16(feats, weight_image_bf16(weight), rb.out_nbr, rb.k_out, rb.kvol,
                             cin, cout)
        k = x.shape[0]
        y = torch.empty((k, cout), dtype=torch.bfloat16 if y_bf16 else torch.float32,
                        device=x.device)
        stats = torch.empty((3 * cout,), dtype=torch.float32, device=x.device)
        ws = _ws(L.lcf_bn_workspace_bytes(k, cout), x.device)
        g = None if gamma is None else gamma.detach().float().contiguous()
        b = None if beta is None else beta.detach().float().contiguous()
        rm = rv = nbt = None
        mom = 0.0
        if run is not None:
            rm, rv, nbt, mom = run
        check(L.lcf_bn_relu_fwd_ex(_ptr(x), 0, k, cout, _ptr(g), _ptr(b), float(eps), int(relu),
                                   _ptr(y), int(bool(y_bf16)), _ptr(stats), _ptr(rm), _ptr(rv),
                                   _ptr(nbt), float(mom), _ptr(ws), ws.numel(), _stream()))
        ctx.save_for_backward(feats, weight, x, y, g, stats)
        ctx.rulebook = rb
        ctx.relu = bool(relu)
        ctx.has = (gamma is not None, beta is not None)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        L = _lib.lib()
        feats, weight, x, y, g, stats = ctx.saved_tensors
        rb = ctx.rulebook
        cin, cout = int(weight.shape[-2]), int(weight.shape[-1])
        k = x.shape[0]
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty((k, cout), dtype=torch.bfloat16, device=x.device)
        dg = torch.empty((cout,), dtype=torch.float32, device=x.device) if ctx.has[0] else None
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if ctx.has[1] else None
        ws = _ws(L.lcf_bn_workspace_bytes(k, cout), x.device)
        check(L.lcf_bn_relu_bwd_ex(_ptr(x), 0, _ptr(y), int(y.dtype == torch.bfloat16), _ptr(dy),
                                   int(dy.dtype == torch.bfloat16), k, cout, _ptr(g), _ptr(stats),
                                   int(ctx.relu), _ptr(dx), 1, _ptr(dg), _ptr(db), _ptr(ws),
                                   ws.numel(), _stream()))
        din = dw = None
        if ctx.needs_input_grad[0]:
            din = conv_forward_b16(dx, weight_image_bf16(weight, transpose=True), rb.in_nbr,
                                   rb.k_in, rb.kvol, cout, cin, out_bf16=True)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad_b16(feats, dx, rb.in_nbr, rb.kvol, cin, cout).view_as(weight).to(
                weight.dtype)
        return din, dw, dg, db, None, None, None, None, None


def conv_bn_relu_train(conv, norm, x, relu, y_bf16):
    """``conv`` (a SparseConvolution whose channels b16_conv_supported) on the SparseConvTensor x
    with bfloat16 features, then nn.BatchNorm1d ``norm`` in training mode (+ ReLU): the fused bf16
    training node; updates the running statistics; returns the new feature rows."""
    rb = conv.get_rulebook(x)
    run = _bn_running(norm)
    y, stats = ConvBNReLUTrainFunction.apply(x.features, conv.weight, norm.weight, norm.bias, rb,
                                             norm.eps, relu, run, y_bf16)
    if norm.track_running_stats and run is None:
        with torch.no_grad():
            k, c = y.shape
            norm.num_batches_tracked += 1
            m = norm.momentum
            if m is None:
                m = 1.0 / float(norm.num_batches_tracked)
            norm.running_mean.mul_(1 - m).add_(stats[:c], alpha=m)
            norm.running_var.mul_(1 - m).add_(stats[c:2 * c], alpha=m * k / max(k - 1, 1))
    return y, rb


# ------------------------------------------------------------------ dense

class DenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch_size, shape):
        x = features.float().contiguous()
        k, c = x.shape
        out = torch.empty((batch_size, c, *shape), dtype=torch.float32, device=x.device)
        check(_lib.lib().lcf_dense(_ptr(x), _ptr(indices), k, None, c, int(batch_size),
                                   i3(shape), 1, _ptr(out), _stream()))
        ctx.save_for_backward(indices)
        ctx.meta = (k, c, int(batch_size), list(shape), features.dtype)
        return out.to(features.dtype)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        k, c, b, shape, dt = ctx.meta
        g = grad.float().contiguous()
        df = torch.empty((k, c), dtype=torch.float32, device=g.device)
        check(_lib.lib().lcf_dense_bwd(_ptr(g), _ptr(indices), k, c, b, i3(shape), _ptr(df),
                                       _stream()))
        return df.to(dt), None, None, None


def dense(features, indices, batch_size, shape):
    """SparseConvTensor.dense(): [B,C,D,H,W]."""
    _need_cuda(features, indices)
    return DenseFunction.apply(features, indices.contiguous(), int(batch_size), to3(list(shape)))


class DenseNHWCBf16Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, indices, batch_size, shape):
        # bf16 rows (the bf16 training path's last layer) go in as they are
        b16 = features.dtype == torch.bfloat16
        x = (features if b16 else features.float()).contiguous()
        k, c = x.shape
        d, h, w = shape
        buf = torch.empty((batch_size, h, w, c * d), dtype=torch.bfloat16, device=x.device)
        check(_lib.lib().lcf_dense_nhwc_bf16_rows(_ptr(x), int(b16), _ptr(indices), k, None, c,
                                                  int(batch_size), i3(shape), 1, _ptr(buf),
                                                  _stream()))
        ctx.save_for_backward(indices)
        ctx.meta = (k, c, int(batch_size), list(shape), features.dtype)
        return buf.permute(0, 3, 1, 2)  # [B, C*D, H, W], channels_last strides

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        k, c, b, shape, dt = ctx.meta
        g = grad.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b16 = dt == torch.bfloat16
        df = torch.empty((k, c), dtype=torch.bfloat16 if b16 else torch.float32, device=g.device)
        check(_lib.lib().lcf_dense_nhwc_bf16_bwd_rows(_ptr(g), _ptr(indices), k, c, b, i3(shape),
                                                      _ptr(df), int(b16), _stream()))
        return df.to(dt), None, None, None


def dense_nhwc_bf16(features, indices, batch_size, shape):
    """SparseConvTensor.dense() + view(N, C*D, H, W) (middle.py:206-209) produced directly as the
    channels_last bf16 tensor the RPN's first convolution reads under bf16 autocast (values =
    the fp32 dense tensor rounded to bf16; the gradient comes back in the same form)."""
    _need_cuda(features, indices)
    return DenseNHWCBf16Function.apply(features, indices.contiguous(), int(batch_size),
                                       to3(list(shape)))


# ------------------------------------------------------------------ VFE streaming kernels

def vfe_mean(voxels, num_points, c):
    _need_cuda(voxels, num_points)
    v = voxels.contiguous()
    k, t, f = v.shape
    out = torch.empty((k, c), dtype=torch.float32, device=v.device)
    check(_lib.lib().lcf_vfe_mean(_ptr(v), _ptr(num_points.int().contiguous()), k, None, t, f,
                                  int(c), _ptr(out), _stream()))
    return out


def vfe_augment(voxels, num_points):
    _need_cuda(voxels, num_points)
    v = voxels.contiguous()
    k, t, f = v.shape
    out = torch.empty((k, t, f + 3), dtype=torch.float32, device=v.device)
    check(_lib.lib().lcf_vfe_augment(_ptr(v), _ptr(num_points.int().contiguous()), k, None, t,
                                     f, _ptr(out), _stream()))
    return out


def vfe_pool(x, num_points, scale, shift, mode):
    _need_cuda(x, num_points)
    x = x.contiguous()
    k, t, u = x.shape
    shape = (k, t, 2 * u) if mode == 0 else (k, u)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    check(_lib.lib().lcf_vfe_pool(_ptr(x), _ptr(num_points.int().contiguous()), k, None, t, u,
                                  _ptr(scale), _ptr(shift), int(mode), _ptr(out), _stream()))
    return out


def vfe_fused(voxels, num_points, lin, k_dev=None, out=None):
    """Whole VoxelFeatureExtractor in one kernel.  lin = [(wimg, cin, cout, scale, shift)] * 3
    (weight images of the three Linear layers + folded BatchNorm).  Returns None when the shape
    is outside what the fused kernel is built for (caller falls back to the layer kernels)."""
    _need_cuda(voxels, num_points)
    v = voxels.contiguous()
    k, t, f = v.shape
    (w1, c0, u1, s1, h1), (w2, _, u2, s2, h2), (w3, _, u3, s3, h3) = lin
    if out is None:
        out = torch.empty((k, u3), dtype=torch.float32, device=v.device)
    rc = _lib.lib().lcf_vfe_fused(_ptr(v), _ptr(num_points), k, _ptr(k_dev), t, f, u1, u2, u3,
                                  _ptr(w1), _ptr(s1), _ptr(h1), _ptr(w2), _ptr(s2), _ptr(h2),
                                  _ptr(w3), _ptr(s3), _ptr(h3), _ptr(out), _stream())
    if rc == -3:
        return None
    check(rc)
    return out


# ------------------------------------------------------------------ BatchNorm1d + ReLU (training)

def bn_relu_supported(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and x.shape[1] % 4 == 0 and x.shape[1] <= 128)


class BNReLUFunction(torch.autograd.Function):
    """y = relu(batch_norm(x)) with batch statistics over the rows of x [K, C] (include/lcf.h:
    lcf_bn_relu_fwd_ex / _bwd_ex).  Returns (y, stats = mean | biased var | inv); with ``run`` =
    (running_mean, running_var, num_batches_tracked, momentum) the forward kernel also updates the
    module's running statistics.  ``y_bf16``: y is stored (and returned) as bf16 -- the input of a
    bf16 MFMA sparse conv, which gathers it -- and the gradient arrives as bf16 too."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, run=None, y_bf16=False):
        L = _lib.lib()
        x = x.contiguous()
        k, c = x.shape
        y = torch.empty((k, c), dtype=torch.bfloat16 if y_bf16 else torch.float32,
                        device=x.device)
        stats = torch.empty((3 * c,), dtype=torch.float32, device=x.device)
        ws = _ws(L.lcf_bn_workspace_bytes(k, c), x.device)
        g = None if gamma is None else gamma.detach().float().contiguous()
        b = None if beta is None else beta.detach().float().contiguous()
        rm = rv = nbt = None
        mom = 0.0
        if run is not None:
            rm, rv, nbt, mom = run
        check(L.lcf_bn_relu_fwd_ex(_ptr(x), 0, k, c, _ptr(g), _ptr(b), float(eps), int(relu),
                                   _ptr(y), int(bool(y_bf16)), _ptr(stats), _ptr(rm), _ptr(rv),
                                   _ptr(nbt), float(mom), _ptr(ws), ws.numel(), _stream()))
        ctx.save_for_backward(x, y, g, stats)
        ctx.relu = bool(relu)
        ctx.has = (gamma is not None, beta is not None)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        L = _lib.lib()
        x, y, g, stats = ctx.saved_tensors
        k, c = x.shape
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty_like(x)
        dg = torch.empty((c,), dtype=torch.float32, device=x.device) if ctx.has[0] else None
        db = torch.empty((c,), dtype=torch.float32, device=x.device) if ctx.has[1] else None
        ws = _ws(L.lcf_bn_workspace_bytes(k, c), x.device)
        check(L.lcf_bn_relu_bwd_ex(_ptr(x), 0, _ptr(y), int(y.dtype == torch.bfloat16), _ptr(dy),
                                   int(dy.dtype == torch.bfloat16), k, c, _ptr(g), _ptr(stats),
                                   int(ctx.relu), _ptr(dx), 0, _ptr(dg), _ptr(db), _ptr(ws),
                                   ws.numel(), _stream()))
        return dx, dg, db, None, None, None, None


class AffineReluSegMaxFunction(torch.autograd.Function):
    """(y, agg) = (relu(x * scale + shift), per-segment max of y on top of ``init``) -- the tail of
    a VFE layer in training (lcf_vfe_affine_relu_segmax_fwd / _bwd).  The gradient of the maximum
    goes to every element that attains it, as voxel_encoder._SegmentMax does."""

    @staticmethod
    def forward(ctx, x, scale, shift, init, seg):
        x = x.contiguous()
        scale, shift = scale.contiguous(), shift.contiguous()
        p, c = x.shape
        y = torch.empty_like(x)
        agg = init.clone(memory_format=torch.contiguous_format)
        check(_lib.lib().lcf_vfe_affine_relu_segmax_fwd(_ptr(x), p, c, _ptr(scale), _ptr(shift),
                                                        _ptr(seg), _ptr(y), _ptr(agg), _stream()))
        ctx.save_for_backward(x, y, agg, scale, init, seg)
        ctx.set_materialize_grads(False)
        return y, agg

    @staticmethod
    def backward(ctx, gy, gagg):
        x, y, agg, scale, init, seg = ctx.saved_tensors
        p, c = x.shape
        if gagg is None:
            gagg = torch.zeros_like(agg)
        gagg = gagg.contiguous()
        gy = None if gy is None else gy.contiguous()
        L = _lib.lib()
        dx = torch.empty_like(x)
        dscale = torch.empty_like(scale)
        dshift = torch.empty_like(scale)
        ws = _ws(L.lcf_bn_workspace_bytes(p, c), x.device)
        check(L.lcf_vfe_affine_relu_segmax_bwd(_ptr(x), _ptr(y), _ptr(agg), _ptr(gy), _ptr(gagg),
                                               _ptr(seg), p, c, _ptr(scale), _ptr(dx),
                                               _ptr(dscale), _ptr(dshift), _ptr(ws), ws.numel(),
                                               _stream()))
        dinit = gagg * (init == agg) if ctx.needs_input_grad[3] else None
        return dx, dscale, dshift, dinit, None


def affine_relu_segmax_supported(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and x.shape[1] % 4 == 0 and x.shape[1] <= 128)


def affine_relu_segmax(x, scale, shift, init, seg):
    """x [p,c], scale / shift [c], init [k,c], seg [p] int32 (voxel of each row) -> (y, agg)."""
    return AffineReluSegMaxFunction.apply(x, scale, shift, init, seg)


class VFELayerFunction(torch.autograd.Function):
    """The tail of a VFE layer in training mode over the real rows only (include/lcf.h:
    lcf_vfe_layer_fwd / _bwd): batch statistics with the padding rows in closed form, BatchNorm
    affine, ReLU, per-voxel max and the (pointwise | repeated) concatenation -- one call each way,
    gradients through the statistics included."""

    @staticmethod
    def forward(ctx, x, xp, wp, seg, gamma, beta, n, k, eps, mode):
        x = x.contiguous()
        p, c = x.shape
        dev = x.device
        L = _lib.lib()
        if xp is not None:
            xp = xp.contiguous()
        agg = torch.empty((k, c), dtype=torch.float32, device=dev)
        out = torch.empty((p, 2 * c), dtype=torch.float32, device=dev) if mode != 2 else None
        stats = torch.empty((5 * c,), dtype=torch.float32, device=dev)
        ws = _ws(L.lcf_vfe_layer_workspace_bytes(p, k, c), dev)
        check(L.lcf_vfe_layer_fwd(_ptr(x), p, _ptr(xp), _ptr(wp), k, c, int(n), _ptr(seg),
                                  _ptr(gamma), _ptr(beta), float(eps), int(mode), _ptr(out),
                                  _ptr(agg), _ptr(stats), _ptr(ws), ws.numel(), _stream()))
        ctx.save_for_backward(x, xp, wp, seg, gamma, agg, stats)
        ctx.meta = (int(n), int(k), int(mode))
        ctx.mark_non_differentiable(stats)
        return (agg if mode == 2 else out), stats

    @staticmethod
    def backward(ctx, gout, _gstats):
        x, xp, wp, seg, gamma, agg, stats = ctx.saved_tensors
        n, k, mode = ctx.meta
        p, c = x.shape
        dev = x.device
        L = _lib.lib()
        gout = gout.contiguous()
        dx = torch.empty_like(x)
        dxp = torch.empty_like(xp) if mode == 0 else None
        dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
        dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
        ws = _ws(L.lcf_vfe_layer_workspace_bytes(p, k, c), dev)
        check(L.lcf_vfe_layer_bwd(_ptr(x), _ptr(agg), _ptr(gout), p, _ptr(xp), _ptr(wp), k, c, n,
                                  _ptr(seg), _ptr(gamma), _ptr(stats), mode, _ptr(dx), _ptr(dxp),
                                  _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), _stream()))
        return dx, dxp, None, None, dgamma, dbeta, None, None, None, None


def vfe_layer_supported(x, norm):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0
            and x.shape[1] % 4 == 0 and x.shape[1] <= 128
            and isinstance(norm, torch.nn.BatchNorm1d) and norm.affine
            and norm.weight.dtype == torch.float32)


def vfe_layer_train(x, xp, wp, seg, norm, n, k, mode):
    """x [p,c] Linear output of the real rows (xp [k,c]: of the padding representatives, mode 0),
    wp [k] padding rows per voxel, seg [p] int32 sorted; nn.BatchNorm1d `norm` in training mode.
    Returns the layer's output ([p,2c], or [k,c] for mode 2) and updates the running statistics
    as nn.BatchNorm1d does."""
    out, stats = VFELayerFunction.apply(x, xp, wp, seg, norm.weight, norm.bias, n, k, norm.eps,
                                        mode)
    if norm.track_running_stats:
        with torch.no_grad():
            c = x.shape[1]
            norm.num_batches_tracked += 1
            m = norm.momentum
            if m is None:
                m = 1.0 / float(norm.num_batches_tracked)
            norm.running_mean.mul_(1 - m).add_(stats[2 * c:3 * c], alpha=m)
            norm.running_var.mul_(1 - m).add_(stats[3 * c:4 * c], alpha=m * n / max(n - 1, 1))
    return out


class BNReLUNHWCFunction(torch.autograd.Function):
    """nn.BatchNorm2d (training) + ReLU on a channels_last bfloat16 activation [N,C,H,W] -- the
    dense RPN under bf16 autocast -- as the row kernels of the sparse path (rows = the N*H*W
    pixels, contiguous in channels_last memory): two passes forward (statistics; normalise +
    ReLU, bf16 out, running statistics updated in the kernel) and two backward, where the library
    route (MIOpen's three-kernel BatchNorm + a separate ReLU / threshold_backward) makes five and
    four passes over the 72 MB tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, run):
        L = _lib.lib()
        n, c, h, w = x.shape
        rows = x.permute(0, 2, 3, 1)  # [N,H,W,C] view of channels_last memory
        if not rows.is_contiguous():
            rows = rows.contiguous()
        k = n * h * w
        y = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=x.device)
        stats = torch.empty((3 * c,), dtype=torch.float32, device=x.device)
        ws = _ws(L.lcf_bn_workspace_bytes(k, c), x.device)
        g = gamma.detach().float().contiguous()
        b = beta.detach().float().contiguous()
        rm = rv = nbt = None
        mom = 0.0
        if run is not None:
            rm, rv, nbt, mom = run
        check(L.lcf_bn_relu_fwd_ex(_ptr(rows), 1, k, c, _ptr(g), _ptr(b), float(eps), 1, _ptr(y),
                                   1, _ptr(stats), _ptr(rm), _ptr(rv), _ptr(nbt), float(mom),
                                   _ptr(ws), ws.numel(), _stream()))
        ctx.save_for_backward(rows, g, b, stats)
        ctx.mark_non_differentiable(stats)
        return y.permute(0, 3, 1, 2), stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        L = _lib.lib()
        rows, g, b, stats = ctx.saved_tensors
        n, h, w, c = rows.shape
        k = n * h * w
        dyr = dy.to(torch.bfloat16).permute(0, 2, 3, 1)
        if not dyr.is_contiguous():
            dyr = dyr.contiguous()
        dev = rows.device
        dx = torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dev)
        dg = torch.empty((c,), dtype=torch.float32, device=dev)
        db = torch.empty((c,), dtype=torch.float32, device=dev)
        ws = _ws(L.lcf_bn_workspace_bytes(k, c), dev)
        check(L.lcf_bn_relu_bwd_b16(_ptr(rows), _ptr(dyr), k, c, _ptr(g), _ptr(b), _ptr(stats), 1,
                                    _ptr(dx), _ptr(dg), _ptr(db), _ptr(ws), ws.numel(),
                                    _stream()))
        return dx.permute(0, 3, 1, 2), dg, db, None, None


def bn_relu_nhwc_supported(norm, x):
    return (isinstance(norm, torch.nn.BatchNorm2d) and norm.training and norm.affine
            and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16
            and x.shape[1] % 8 == 0 and x.shape[1] <= 128 and x.numel() > 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and norm.weight.dtype == torch.float32 and _bn_running(norm) is not None)


def bn_relu_train_nhwc(norm, x):
    """relu(norm(x)) for a training-mode nn.BatchNorm2d on a channels_last bfloat16 x (see
    BNReLUNHWCFunction); running statistics updated as the module would."""
    y, _ = BNReLUNHWCFunction.apply(x, norm.weight, norm.bias, norm.eps, _bn_running(norm))
    return y


def _bn_running(norm):
    """(running_mean, running_var, num_batches_tracked, momentum) for the kernels that update the
    running statistics themselves, or None (no tracking, cumulative averaging, non-fp32 buffers:
    the caller then updates them with torch ops)."""
    if not norm.track_running_stats or norm.momentum is None or norm.running_mean is None:
        return None
    rm, rv, nbt = norm.running_mean, norm.running_var, norm.num_batches_tracked
    if rm.dtype != torch.float32 or rv.dtype != torch.float32 or not rm.is_cuda or \
            not rm.is_contiguous() or not rv.is_contiguous():
        return None
    if nbt is not None and nbt.dtype != torch.int64:
        return None
    return rm, rv, nbt, float(norm.momentum)


def bn_relu_train(norm, x, relu=True, y_bf16=False):
    """nn.BatchNorm1d `norm` in training mode (+ ReLU) on x [K, C]: the fused kernels, including
    the running-statistics update nn.BatchNorm1d would have done (unbiased variance, momentum)."""
    run = _bn_running(norm)
    y, stats = BNReLUFunction.apply(x, norm.weight, norm.bias, norm.eps, relu, run, y_bf16)
    if norm.track_running_stats and run is None:
        with torch.no_grad():
            k, c = x.shape
            norm.num_batches_tracked += 1
            m = norm.momentum
            if m is None:
                m = 1.0 / float(norm.num_batches_tracked)
            norm.running_mean.mul_(1 - m).add_(stats[:c], alpha=m)
            norm.running_var.mul_(1 - m).add_(stats[c:2 * c], alpha=m * k / max(k - 1, 1))
    return y
