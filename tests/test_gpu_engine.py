"""GPU: the graph-captured FrameEngine (no host sync, device-side row counts) reproduces the
module-by-module path bit for bit, across frames of different sizes, and reports overflow."""
import contextlib

import pytest
import torch

from objectdetection_amd import _lib, synth
from objectdetection_amd.frontend import SparseFrontEnd, randomize_

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def one_wave_set():
    """The forward picks one or two wave sets per layer from the row count it expects (the
    engine: calibrated, frozen in the graph; the module path: the frame's own), and the two sum
    the kernel offsets in a different order.  Pinned to one set, both paths run the same
    kernels and must agree bit for bit."""
    L = _lib.lib()
    L.lcf_set_option(_lib.OPT_CONV_SETS, 1)
    try:
        yield
    finally:
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)


def _clouds(cuda, seeds, n):
    return [torch.from_numpy(synth.vkitti_frame(s, n)).to(cuda) for s in seeds]


@pytest.mark.parametrize("name", ["M1", "M2"])
def test_engine_matches_eager(cuda, name):
    model = randomize_(SparseFrontEnd(name).eval(), 0).to(cuda)
    calib = _clouds(cuda, [0, 1], 20000)
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=calib)
        # frames of different sizes through the SAME captured graph, in arbitrary order
        for seed, n in [(0, 20000), (5, 12000), (1, 20000), (7, 3000), (5, 12000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            out = eng.forward(c).clone()
            eng.check()
            ref = model([c])
            assert out.shape == ref.shape
            if name == "M1":  # same kernels in the same order: bit-identical
                assert torch.equal(out, ref)
            else:  # M2: VFE Linear layers on the fp32-MFMA linear kernel vs a library GEMM
                err = (out - ref).abs().max() / ref.abs().max()
                assert err.item() < 1e-4  # the north-star fp32 feature tolerance
                assert torch.equal(out != 0, ref != 0) or err.item() < 1e-5
        ks = eng.counts()
        assert ks[0] > 0 and len(ks) == 1 + eng.n_strided


def test_engine_without_graph_and_empty_frame(cuda):
    model = randomize_(SparseFrontEnd("M1").eval(), 0).to(cuda)
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=_clouds(cuda, [0], 8000), use_graph=False)
        c = _clouds(cuda, [3], 8000)[0]
        assert torch.equal(eng.forward(c), model([c]))
        empty = torch.zeros((0, 4), device=cuda)
        out = eng.forward(empty)
        assert eng.counts()[0] == 0
        assert float(out.abs().max()) == 0.0


def test_engine_overflow_is_reported_and_replanned(cuda):
    model = randomize_(SparseFrontEnd("M1").eval(), 0).to(cuda)
    small = _clouds(cuda, [0], 1500)
    big = _clouds(cuda, [1], 30000)[0]
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=small, max_points=40000, margin=1.0)
        eng.forward(big)
        assert eng.overflowed()
        with pytest.raises(RuntimeError):
            eng.check()
        out = eng.forward(big, sync=True)  # re-plans with larger capacities
        assert not eng.overflowed()
        assert torch.equal(out, model([big]))


def test_engine_dense_120k_point_frame(cuda):
    """BASELINE.json configs[4] size: a ~120k-point SYNTHIA-shaped frame (hash occupancy stress)
    through the captured engine equals the module path bit for bit (M1: same kernels)."""
    model = randomize_(SparseFrontEnd("M1", max_voxels=160000).eval(), 0).to(cuda)
    c = torch.from_numpy(synth.synthia_frame(0, 120000)).to(cuda)
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=[c])
        out = eng.forward(c).clone()
        eng.check()
        ks = eng.counts()
        ref = model([c])
    assert ks[0] > 60000  # tens of thousands of voxels per level
    assert torch.equal(out, ref)


@pytest.mark.parametrize("name", ["M1", "M2"])
def test_engine_default_kernel_selection(cuda, name):
    """Default selection (two wave sets where the calibrated row counts say every tile is
    resident): engine and module path may then sum the offsets of a layer in a different order;
    they agree to fp32 reassociation error, far inside the 1e-4 feature tolerance."""
    model = randomize_(SparseFrontEnd(name).eval(), 0).to(cuda)
    with torch.no_grad():
        eng = model.engine(calib=_clouds(cuda, [0, 1], 20000))
        for seed, n in [(0, 20000), (7, 3000), (2, 20000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            out = eng.forward(c).clone()
            eng.check()
            ref = model([c])
            err = (out - ref).abs().max() / ref.abs().max()
            assert err.item() < (1e-5 if name == "M1" else 1e-4)
            again = eng.forward(c)
            assert torch.equal(out, again)  # each path by itself is run-to-run deterministic


def test_engine_append_keeps_the_cloud_on_the_device(cuda):
    """Light-curtain loop: curtain returns are appended to the cloud that already sits in the
    engine (forward_append copies only the new points); each step equals a fresh forward of the
    concatenated cloud."""
    model = randomize_(SparseFrontEnd("M1").eval(), 0).to(cuda)
    base = _clouds(cuda, [0], 12000)[0]
    extra = [torch.from_numpy(synth.vkitti_frame(20 + i, 3000)).to(cuda) for i in range(3)]
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=[torch.cat([base] + extra)], max_points=40000)
        out = eng.forward(base).clone()
        assert torch.equal(out, model([base]))
        cloud = base
        for e in extra:
            cloud = torch.cat([cloud, e])
            out = eng.forward_append(e).clone()
            eng.check()
            assert torch.equal(out, model([cloud]))
        with pytest.raises(RuntimeError):
            eng.forward_append(torch.zeros((40000, 4), device=cuda))


@pytest.mark.parametrize("name", ["M1", "M2"])
def test_engine_bf16x3_stays_inside_the_fp32_feature_tolerance(cuda, name):
    """conv_math="bf16x3": the 32/64/128-channel layers of the whole stack on split bf16 operands;
    the network's output against the fp32 module path, relative to its max magnitude, stays
    inside the north-star 1e-4 (measured ~1e-5), same active cells, deterministic replays."""
    model = randomize_(SparseFrontEnd(name).eval(), 0).to(cuda)
    with torch.no_grad():
        eng = model.engine(calib=_clouds(cuda, [0, 1], 20000), conv_math="bf16x3")
        assert any(l.x3 for l in eng.layers) and not all(l.x3 for l in eng.layers)
        for seed, n in [(0, 20000), (7, 3000), (2, 20000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            out = eng.forward(c).clone()
            eng.check()
            ref = model([c])
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-4
            assert err < 3e-5
            assert torch.equal(eng.forward(c), out)
    with pytest.raises(ValueError):
        model.engine(calib=_clouds(cuda, [0], 3000), conv_math="bf16")


def test_module_path_bf16x3_equals_engine_bf16x3(cuda):
    """ops.conv_math("bf16x3") puts the module-by-module (spconv API) inference path on the same
    split-operand kernels as FrameEngine(conv_math="bf16x3"): same kernels pinned to one wave set
    -> bit-identical; inside the fp32 tolerance of the fp32 module path."""
    from objectdetection_amd import ops
    model = randomize_(SparseFrontEnd("M1").eval(), 0).to(cuda)
    c = _clouds(cuda, [4], 15000)[0]
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=[c], conv_math="bf16x3")
        out = eng.forward(c).clone()
        ref32 = model([c])
        with ops.conv_math("bf16x3"):
            ref3 = model([c])
        assert ops.get_conv_math() == "f32"
    assert torch.equal(out, ref3)
    err = ((ref3 - ref32).abs().max() / ref32.abs().max()).item()
    assert 0 < err < 1e-4


def test_engine_bf16x8_is_fp32_grade(cuda):
    """conv_math="bf16x8": the 64-channel layers on three exact bf16 planes; the network's output
    stays at fp32 reassociation distance from the fp32 module path."""
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(cuda)
    with torch.no_grad():
        eng = model.engine(calib=_clouds(cuda, [0, 1], 20000), conv_math="bf16x8")
        assert any(l.x3 and l.planes == 3 for l in eng.layers)
        for seed, n in [(0, 20000), (7, 3000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            out = eng.forward(c).clone()
            eng.check()
            ref = model([c])
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-6
            assert torch.equal(eng.forward(c), out)


@pytest.mark.parametrize("mode", ["auto", "all"])
def test_engine_bf16x6_is_fp32_grade(cuda, mode, monkeypatch):
    """conv_math="bf16x6": rows travel between the chosen layers as three exact bf16 planes
    (lcf_split_planes at the entry, lcf_indice_conv_fwd_x6 writing planes or fp32 rows); the
    network's output stays at fp32 reassociation distance from the fp32 module path, frames of
    other sizes through the same graph included, run to run bit-identical.  "auto" = the layers
    where the form measured ahead (32 -> 32 on this frame), "all" = every candidate layer."""
    monkeypatch.setenv("LCF_ENGINE_X6", mode)
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(cuda)
    with torch.no_grad():
        eng = model.engine(calib=_clouds(cuda, [0, 1], 20000), conv_math="bf16x6")
        used = [l for l in eng.layers if l.use_x6]
        assert used and all(not l.planned for l in used)
        assert any(l.split_in for l in used) and any(not l.out_planes for l in used)
        if mode == "all":
            assert sum(1 for l in used if l.out_planes) >= 6
        for seed, n in [(0, 20000), (7, 3000), (3, 24000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            out = eng.forward(c).clone()
            eng.check()
            ref = model([c])
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-6, err
            assert torch.equal(eng.forward(c), out)


def test_load_points_copies_and_counts_in_one_launch(cuda):
    """lcf_load_points: the rows land where dst points, the device-side counter gets n_total;
    what the float4 copy cannot do is refused (the engine then copies and fills separately)."""
    L = _lib.lib()
    src = torch.randn((1000, 4), device=cuda)
    buf = torch.zeros((3000, 4), device=cuda)
    n_dev = torch.zeros((1,), dtype=torch.int32, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    assert L.lcf_load_points(src.data_ptr(), 1000, 4, buf[500:].data_ptr(), 1500,
                             n_dev.data_ptr(), st) == 0
    assert torch.equal(buf[500:1500], src) and int(n_dev) == 1500
    assert float(buf[:500].abs().sum()) == 0 and float(buf[1500:].abs().sum()) == 0
    assert L.lcf_load_points(src.data_ptr(), 0, 4, buf.data_ptr(), 7, n_dev.data_ptr(), st) == 0
    assert int(n_dev) == 7
    assert L.lcf_load_points(src.data_ptr(), 1000, 4, buf.data_ptr(), 999, n_dev.data_ptr(),
                             st) == -1  # n_total < n
    assert L.lcf_load_points(src.data_ptr(), 5, 3, buf.data_ptr(), 5, n_dev.data_ptr(),
                             st) == -3  # 15 floats: not a whole number of float4
    assert L.lcf_load_points(src.data_ptr() + 4, 4, 4, buf.data_ptr(), 4, n_dev.data_ptr(),
                             st) == -3  # misaligned source
    torch.cuda.synchronize()


def test_batched_fills_run_as_one_launch_with_the_same_result(cuda):
    """lcf_ctx_fill_batch_begin / _flush: fills recorded in a caller-owned context (ragged sizes,
    unaligned heads, more segments than one plan holds) land exactly as separate lcf_fill_u32
    calls would; fills made meanwhile WITHOUT that context (or with another one) run at once."""
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    sizes = [1, 3, 4, 5, 255, 256, 1000, 4097, 70001, 12, 7, 33, 64, 129, 100000, 2, 9, 31, 513]
    bufs = [torch.full((n + 3,), 5, dtype=torch.int32, device=cuda) for n in sizes]
    ctx, other = _lib.Context(), _lib.Context()
    ctx.fill_batch_begin()
    for i, (b, n) in enumerate(zip(bufs, sizes)):
        off = i % 3  # unaligned starts
        assert L.lcf_fill_u32(b[off:].data_ptr(), 1000 + i, n * 4, st, ctx.handle) == 0
    bystander = torch.zeros(8, dtype=torch.int32, device=cuda)
    assert L.lcf_fill_u32(bystander.data_ptr(), 3, 16, st) == 0               # no context
    assert L.lcf_fill_u32(bystander[4:].data_ptr(), 4, 16, st, other.handle) == 0  # another one
    torch.cuda.synchronize()
    assert bystander.tolist() == [3, 3, 3, 3, 4, 4, 4, 4]
    assert all(int((b != 5).sum()) == 0 for b in bufs[-3:])  # recorded, not yet run
    ctx.fill_batch_flush(st)
    for i, (b, n) in enumerate(zip(bufs, sizes)):
        off = i % 3
        ref = torch.full_like(b, 5)
        ref[off:off + n] = 1000 + i
        assert torch.equal(b, ref), i
    # outside a batch fills run immediately again
    assert L.lcf_fill_u32(bufs[0].data_ptr(), 1, 4, st) == 0
    torch.cuda.synchronize()
    assert int(bufs[0][0]) == 1


def test_dense_clear_rows_through_the_c_abi(cuda):
    """lcf_dense_clear_rows: the cells the previous frame's rows were scattered
    to are zeroed, `prev_coors` becomes this frame's coordinates (-1 beyond the count); alone
    and as the one clear record of a fill batch (behind recorded fills), ragged sizes."""
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    C, D, H, W, cap = 64, 2, 20, 17, 301
    shape = _lib.i3((D, H, W))
    g = torch.Generator().manual_seed(3)

    def frame(k):
        cells = torch.randperm(D * H * W, generator=g)[:k]
        co = torch.zeros((cap, 4), dtype=torch.int32)
        co[:k, 1], co[:k, 2], co[:k, 3] = cells // (H * W), cells // W % H, cells % W
        co[k:] = 12345  # stale rows beyond the count: never read as coordinates
        return co.to(cuda)

    def scatter(dense, co, k):
        idx = co[:k].long()
        dense[0, :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.rand((C, k), device=cuda) + 1.0

    for batched in (False, True):
        dense = torch.zeros((1, C, D, H, W), device=cuda)
        prev = torch.full((cap, 4), -1, dtype=torch.int32, device=cuda)
        k_dev = torch.zeros(1, dtype=torch.int32, device=cuda)
        ctx = _lib.Context()
        side = torch.full((1000,), 7, dtype=torch.int32, device=cuda)
        for k in (200, 301, 0, 5, 150):
            co = frame(k)
            k_dev.fill_(k)
            if batched:
                ctx.fill_batch_begin()
                assert L.lcf_fill_u32(side.data_ptr(), k, 4000, st, ctx.handle) == 0
            assert L.lcf_dense_clear_rows(dense.data_ptr(), prev.data_ptr(), co.data_ptr(), cap,
                                          k_dev.data_ptr(), C, 1, shape, st,
                                          ctx.handle if batched else None) == 0
            if batched:
                # a second record in one batch is refused
                assert L.lcf_dense_clear_rows(dense.data_ptr(), prev.data_ptr(), co.data_ptr(),
                                              cap, k_dev.data_ptr(), C, 1, shape, st,
                                              ctx.handle) == -3
                ctx.fill_batch_flush(st)
                assert int((side != k).sum()) == 0
            assert float(dense.abs().sum()) == 0.0  # everything the last frame wrote is gone
            assert torch.equal(prev[:k], co[:k]) and int((prev[k:] != -1).sum()) == 0
            scatter(dense, co, k)
    assert L.lcf_dense_clear_rows(None, prev.data_ptr(), co.data_ptr(), cap, None, C, 1, shape,
                                  st, None) == -1


def test_engine_clears_only_the_rows_of_the_previous_frame(cuda):
    """The engine's dense output equals the whole-buffer-fill engine's bit for bit over frames of
    different sizes (a small frame after a large one leaves no stale cell), through the graph."""
    model = randomize_(SparseFrontEnd("M1").eval(), 0).to(cuda)
    calib = _clouds(cuda, [0, 1], 20000)
    with torch.no_grad(), one_wave_set():
        eng = model.engine(calib=calib, dense_clear="rows")
        assert eng.clear_rows
        full = model.engine(calib=calib)
        assert not full.clear_rows
        for seed, n in [(0, 20000), (7, 3000), (1, 20000), (9, 100), (5, 12000), (5, 12000)]:
            c = torch.from_numpy(synth.vkitti_frame(seed, n)).to(cuda)
            a, b = eng.forward(c).clone(), full.forward(c).clone()
            eng.check()
            assert torch.equal(a, b), (seed, n)
        empty = torch.zeros((0, 4), device=cuda)
        assert float(eng.forward(empty).abs().sum()) == 0.0
        assert eng.scan_fallbacks() == 0


def test_two_engines_enqueue_from_two_threads(cuda):
    """Re-entrancy (SURVEY.md section 8b "Threading"): each engine owns its lcf_ctx, so two engines
    -- eager enqueue, i.e. ~60 option-dependent C calls per frame each -- driven from two threads
    at once, beside a third thread that runs plain self-resetting ops.* calls on fresh workspaces,
    all give what they give alone."""
    import threading
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(cuda)
    clouds = {0: _clouds(cuda, [11, 12, 13], 9000), 1: _clouds(cuda, [21, 22, 23], 14000)}
    with torch.no_grad():
        want = {t: [model([c]).clone() for c in cs] for t, cs in clouds.items()}
        engines = {t: model.engine(calib=cs, use_graph=False) for t, cs in clouds.items()}
        vox_ref = [ops_voxelize(c) for c in clouds[0]]
    torch.cuda.synchronize()
    errs, start = [], threading.Barrier(3)

    def drive(t):
        try:
            torch.cuda.set_device(cuda)
            s = torch.cuda.Stream(device=cuda)
            start.wait()
            with torch.no_grad(), torch.cuda.stream(s):
                for rep in range(6):
                    for c, w in zip(clouds[t], want[t]):
                        out = engines[t].forward(c)
                        s.synchronize()
                        engines[t].check()
                        err = ((out - w).abs().max() / w.abs().max()).item()
                        assert err < 1e-5, (t, rep, err)
        except BaseException as e:  # noqa: BLE001 - re-raised in the main thread
            errs.append(e)

    def bystander():
        try:
            torch.cuda.set_device(cuda)
            s = torch.cuda.Stream(device=cuda)
            start.wait()
            with torch.cuda.stream(s):
                for rep in range(12):
                    for c, ref in zip(clouds[0], vox_ref):
                        got = ops_voxelize(c)  # fresh torch.empty workspace: relies on the reset
                        s.synchronize()
                        assert all(torch.equal(a, b) for a, b in zip(got, ref))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=drive, args=(0,)), threading.Thread(target=drive, args=(1,)),
           threading.Thread(target=bystander)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=110)
    assert not any(th.is_alive() for th in ths)
    if errs:
        raise errs[0]


def ops_voxelize(c):
    from objectdetection_amd import ops
    return ops.voxelize(c, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000)


def test_overflow_replan_is_bounded_and_uses_observed_counts(cuda):
    """forward(sync=True) on a frame that exceeds the calibrated capacities: one measuring pass
    with loose capacities, a tight re-plan from the observed counts, one re-run -- no recursion,
    and the kernel-selection hints follow."""
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(cuda)
    small = _clouds(cuda, [1], 4000)
    big = _clouds(cuda, [2], 30000)[0]
    with torch.no_grad():
        eng = model.engine(calib=small, margin=1.0)
        before = list(eng.level_caps)
        out = eng.forward(big, sync=True).clone()
        ref = model([big])
    assert not eng.overflowed()
    assert all(a >= b for a, b in zip(eng.level_caps, before)) and eng.level_caps != before
    assert eng.level_need[0] > 20000
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.gpu
def test_frame_pipeline_gives_each_frame_the_single_engine_result(cuda):
    """FramePipeline: frames dealt over two engines on two streams come out as the one-engine
    path computes them, bit for bit, whatever is in flight beside them."""
    from objectdetection_amd.engine import FramePipeline
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(cuda)
    calib = _clouds(cuda, [0, 1], 20000)
    with torch.no_grad():
        eng = model.engine(calib=calib)
        pipe = FramePipeline(model, calib=calib, depth=2)
        frames = [torch.from_numpy(synth.vkitti_frame(s, n)).to(cuda)
                  for s, n in [(3, 20000), (4, 9000), (5, 20000), (6, 15000), (3, 20000)]]
        torch.cuda.synchronize()
        outs = []
        for c in frames:
            out, done = pipe.submit(c)
            outs.append((out, done))
            if len(outs) >= 2:  # consume a result before its slot is used again
                o, ev = outs[-2]
                ev.synchronize()
                outs[-2] = o.clone()
        pipe.synchronize()
        outs[-1] = outs[-1][0].clone()
        pipe.check()
        for c, o in zip(frames, outs):
            ref = eng.forward(c).clone()
            eng.check()
            assert torch.equal(o, ref)


@pytest.mark.gpu
def test_frame_batch_rows_are_the_models_batch_rows():
    """FrameBatch (B engines writing the B slices of one dense buffer) against the eager module
    on the same batch of clouds, and row b against a single FrameEngine on cloud b."""
    from objectdetection_amd.engine import FrameBatch, FrameEngine
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 3).to(dev)
    clouds = [torch.from_numpy(synth.vkitti_frame(i, n)).to(dev) for i, n in ((0, 9000), (1, 6000))]
    eng = FrameBatch(model, batch=2, calib=clouds)
    out = eng(clouds).clone()
    eng.check()
    with torch.no_grad():
        ref = model(clouds)
    assert out.shape == ref.shape and out.shape[0] == 2
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 1e-4 * scale
    single = FrameEngine(model, calib=clouds)
    for b, c in enumerate(clouds):
        assert torch.equal(single(c), out[b:b + 1])
    # a second call, clouds swapped: each slot re-arms its own workspaces
    out2 = eng(clouds[::-1]).clone()
    assert torch.equal(out2[0], out[1]) and torch.equal(out2[1], out[0])
    with pytest.raises(ValueError):
        eng(clouds[:1])
