"""GPU parity: indice_conv forward / dgrad / wgrad, dense, VFE kernels vs the CPU oracle."""
import ctypes

import numpy as np
import pytest
import torch

from objectdetection_amd import ops, synth

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north-star tolerance for fp32 features (relative to the tensor's max magnitude)
BF16_RTOL = 2e-2  # stated bf16 tolerance (north star): bf16 operands, fp32 accumulation


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _frame_coors(oracle, seed=0, n=20000):
    _, c, _ = oracle.voxelize(synth.vkitti_frame(seed, n), synth.VOXEL_SIZE, synth.PC_RANGE, 5,
                              40000)
    return np.ascontiguousarray(np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1))


@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64),
                                      (128, 16), (3, 8), (5, 24), (64, 128), (20, 40)])
@pytest.mark.parametrize("subm", [True, False])
def test_forward_matches_oracle(oracle, cuda, cin, cout, subm):
    rng = np.random.default_rng(cin * 131 + cout)
    coors = _frame_coors(oracle, 0, 6000)
    # densify a little so that offsets other than the centre carry pairs
    extra = coors.copy()
    extra[:, 3] = np.minimum(extra[:, 3] + 1, 1407)
    coors = np.unique(np.concatenate([coors, extra], 0), axis=0)
    rng.shuffle(coors, axis=0)
    coors = np.ascontiguousarray(coors)
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    assert rb.k_out == k_out
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    out = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, rb.k_out, 27, cin, cout)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    # fused epilogue: scale/shift/relu
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    out2 = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, rb.k_out, 27, cin, cout,
                            torch.from_numpy(sc).to(cuda), torch.from_numpy(sh).to(cuda), True)
    ref2 = np.maximum(ref * sc[None] + sh[None], 0)
    assert rel_err(out2.cpu().numpy(), ref2) < RTOL


@pytest.mark.parametrize("impl", [3, 4])
@pytest.mark.parametrize("cin,cout,subm", [(128, 16, True), (16, 16, True), (16, 32, False),
                                           (4, 16, True), (5, 24, False), (32, 32, True)])
def test_thin_layer_forms_match_oracle(oracle, cuda, cin, cout, subm, impl):
    """The split-slab (3) and thin-tile (4) forwards on the shapes they serve, pinned by
    LCF_OPT_CONV_IMPL in a context: same results as the oracle, fused epilogue included, and
    run-to-run bit-identical."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(cin * 7 + cout + impl)
    coors = _frame_coors(oracle, 2, 9000)
    extra = coors.copy()
    extra[:, 3] = np.minimum(extra[:, 3] + 1, 1407)
    coors = np.unique(np.concatenate([coors, extra[::3]], 0), axis=0)
    rng.shuffle(coors, axis=0)
    coors = np.ascontiguousarray(coors)
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt, wimg = torch.from_numpy(x).to(cuda), ops.weight_image(torch.from_numpy(w).to(cuda))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(cuda)
    sh = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(cuda)
    ctx = _lib.Context()
    ctx.set(_lib.OPT_CONV_IMPL, impl)
    L = _lib.lib()

    def run(scale, shift, relu):
        out = torch.empty((k_out, cout), dtype=torch.float32, device=cuda)
        P = lambda t: None if t is None else t.data_ptr()
        _lib.check(L.lcf_indice_conv_fwd(P(xt), P(wimg), P(rb.out_nbr), k_out, None, 27, cin, cout,
                                         P(scale), P(shift), int(relu), P(out),
                                         torch.cuda.current_stream().cuda_stream, ctx.handle))
        return out
    a, b = run(None, None, False), run(None, None, False)
    assert torch.equal(a, b)
    assert rel_err(a.cpu().numpy(), ref) < RTOL
    ref2 = np.maximum(ref * sc.cpu().numpy()[None] + sh.cpu().numpy()[None], 0)
    assert rel_err(run(sc, sh, True).cpu().numpy(), ref2) < RTOL


@pytest.mark.parametrize("cin,cout,subm", [(16, 32, False), (32, 32, True), (64, 64, True),
                                           (4, 16, True), (128, 16, True), (5, 24, False),
                                           (128, 128, True)])
def test_backward_matches_oracle(oracle, cuda, cin, cout, subm):
    rng = np.random.default_rng(7 + cin)
    coors = _frame_coors(oracle, 1, 5000)
    extra = coors.copy()
    extra[:, 2] = np.minimum(extra[:, 2] + 1, 1599)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    g = rng.standard_normal((k_out, cout)).astype(np.float32)
    din_ref, dw_ref = oracle.indice_conv_bwd(x, w, g, pairs, pn)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    wt = torch.from_numpy(w).to(cuda).requires_grad_(True)
    out = ops.indice_conv(xt, wt, rb)
    out.backward(torch.from_numpy(g).to(cuda))
    assert rel_err(xt.grad.cpu().numpy(), din_ref) < RTOL
    assert rel_err(wt.grad.cpu().numpy(), dw_ref) < RTOL


@pytest.mark.parametrize("cin,cout,subm", [(32, 32, True), (64, 64, True), (32, 64, False),
                                           (64, 32, True), (32, 128, False), (128, 32, True)])
def test_matrix_core_wgrad_matches_oracle(oracle, cuda, cin, cout, subm):
    """lcf_indice_conv_wgrad_bf16: split operands (planes = 2) inside the fp32 tolerance, plain
    bf16 operands (planes = 1) inside the stated bf16 tolerance, both against the oracle; rows
    with magnitudes over several octaves; blocks with more than 32 pairs per 64 rows (two steps)
    and with none."""
    rng = np.random.default_rng(3 * cin + cout)
    coors = _frame_coors(oracle, 6, 5000)
    extra = coors.copy()
    extra[:, 3] = np.minimum(extra[:, 3] + 1, 1407)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = (rng.standard_normal((k, cin)) * np.exp2(rng.integers(-4, 5, (k, 1)))).astype(np.float32)
    w = np.zeros((3, 3, 3, cin, cout), np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    g = (rng.standard_normal((k_out, cout)) *
         np.exp2(rng.integers(-4, 5, (k_out, 1)))).astype(np.float32)
    _, dw_ref = oracle.indice_conv_bwd(x, w, g, pairs, pn)
    dw_ref = dw_ref.reshape(27, cin, cout)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt, gt = torch.from_numpy(x).to(cuda), torch.from_numpy(g).to(cuda)
    assert ops.wgrad_mc_supported(cin, cout)
    d2 = ops.conv_wgrad(xt, gt, rb.in_nbr, 27, cin, cout, planes=2).cpu().numpy()
    d1 = ops.conv_wgrad(xt, gt, rb.in_nbr, 27, cin, cout, planes=1).cpu().numpy()
    d0 = ops.conv_wgrad(xt, gt, rb.in_nbr, 27, cin, cout).cpu().numpy()
    assert rel_err(d0, dw_ref) < RTOL
    assert rel_err(d2, dw_ref) < RTOL
    assert rel_err(d2, dw_ref) < 2e-5
    assert rel_err(d1, dw_ref) < BF16_RTOL
    # per offset too: an offset with few pairs must not hide behind the centre's magnitude
    for o in range(27):
        if np.abs(dw_ref[o]).max() > 0:
            assert rel_err(d2[o], dw_ref[o]) < RTOL


def test_matrix_core_wgrad_rejects_unsupported_shapes(cuda):
    assert not ops.wgrad_mc_supported(16, 32) and not ops.wgrad_mc_supported(128, 64)
    t = torch.zeros((8, 16), device=cuda)
    n = torch.full((8, 27), -1, dtype=torch.int32, device=cuda)
    with pytest.raises(RuntimeError):
        ops.conv_wgrad(t, torch.zeros((8, 32), device=cuda), n, 27, 16, 32, planes=2)


@pytest.mark.parametrize("cin,cout", [(7, 16), (16, 7), (32, 64), (64, 32), (128, 128),
                                      (128, 64), (100, 120), (3, 5)])
def test_linear_wgrad_equals_float64_product(cuda, cin, cout):
    """lcf_linear_wgrad (weight gradient of the VFE's Linear layers: x^T dy over ~10^5 dense
    rows) against the float64 product, within the fp32 tolerance; ragged row counts; through
    TallLinearFunction: same outputs and gradients as nn.Linear."""
    rng = np.random.default_rng(cin * 7 + cout)
    for n in (1, 63, 257, 40001):
        x = rng.standard_normal((n, cin)).astype(np.float32)
        dy = rng.standard_normal((n, cout)).astype(np.float32)
        ref = x.astype(np.float64).T @ dy.astype(np.float64)
        got = ops.linear_wgrad(torch.from_numpy(x).to(cuda), torch.from_numpy(dy).to(cuda))
        assert got.shape == (cin, cout)
        assert rel_err(got.cpu().numpy(), ref) < RTOL
    assert ops.linear_wgrad(torch.zeros((0, cin), device=cuda),
                            torch.zeros((0, cout), device=cuda)).abs().sum().item() == 0
    lin = torch.nn.Linear(cin, cout, bias=(cin % 2 == 0)).to(cuda)
    xt = torch.from_numpy(x).to(cuda)
    xa = xt.clone().requires_grad_(True)
    xb = xt.clone().requires_grad_(True)
    g = torch.from_numpy(dy).to(cuda)
    ya = ops.tall_linear(lin, xa)
    ya.backward(g)
    ga = [p.grad.clone() for p in lin.parameters()]
    lin.zero_grad()
    yb = lin(xb)
    yb.backward(g)
    assert torch.equal(ya, yb)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    for a, p in zip(ga, lin.parameters()):
        assert rel_err(a.cpu().numpy(), p.grad.cpu().numpy()) < RTOL


@pytest.mark.parametrize("c", [16, 64, 128])
def test_vfe_affine_relu_segmax_equals_torch_composition(cuda, c):
    """lcf_vfe_affine_relu_segmax_fwd / _bwd (the tail of a VFE layer in training) against
    relu(x * scale + shift) followed by voxel_encoder._SegmentMax in torch: outputs bit-identical
    (same fp32 operations; max is order-independent), gradients of x / scale / shift / init to
    fp32 summation order; voxels whose initial value wins; gy absent (the final layer)."""
    from objectdetection_amd.voxel_encoder import _SegmentMax
    g = torch.Generator(device="cuda").manual_seed(c)
    k, p = 3000, 7001
    seg = torch.sort(torch.randint(0, k, (p,), device=cuda, generator=g)).values
    x = torch.randn((p, c), device=cuda, generator=g)
    scale = torch.rand((c,), device=cuda, generator=g) + 0.5
    shift = torch.randn((c,), device=cuda, generator=g) * 0.3
    pad = torch.relu(torch.randn((k, c), device=cuda, generator=g))
    has = torch.rand((k, 1), device=cuda, generator=g) < 0.5
    init = torch.where(has, pad, torch.full_like(pad, torch.finfo(torch.float32).min))
    gy = torch.randn((p, c), device=cuda, generator=g)
    ga = torch.randn((k, c), device=cuda, generator=g)
    for with_gy in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (x, scale, shift, init)]
        ref_leaves = [t.clone().requires_grad_(True) for t in (x, scale, shift, init)]
        y, agg = ops.affine_relu_segmax(*leaves, seg.int())
        yr = torch.relu(ref_leaves[0] * ref_leaves[1] + ref_leaves[2])
        aggr = _SegmentMax.apply(ref_leaves[3], seg, yr)
        assert torch.equal(y, yr) and torch.equal(agg, aggr)
        if with_gy:
            torch.autograd.backward([y, agg], [gy, ga])
            torch.autograd.backward([yr, aggr], [gy, ga])
        else:
            agg.backward(ga)
            aggr.backward(ga)
        for a, b in zip(leaves, ref_leaves):
            assert rel_err(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 1e-5


def test_linearity_and_determinism_full_size(oracle, cuda):
    """Size-independent properties at the full BASELINE size (K ~ 40k rows, 64 channels)."""
    coors = _frame_coors(oracle, 0, 20000)
    shape = [41, 1600, 1408]
    ind = torch.from_numpy(coors).to(cuda)
    rb = ops.build_rulebook(ind, 1, shape, 3, 2, 1, 1, False)
    rb2 = ops.build_rulebook(rb.out_indices, 1, rb.out_shape, 3, 1, 0, 1, True)
    g = torch.Generator(device="cuda").manual_seed(0)
    k = rb2.k_in
    a = torch.randn((k, 64), device=cuda, generator=g)
    b = torch.randn((k, 64), device=cuda, generator=g)
    w = torch.randn((27, 64, 64), device=cuda, generator=g) / 40
    img = ops.weight_image(w)
    f = lambda t: ops.conv_forward(t, img, rb2.out_nbr, k, 27, 64, 64)
    ya, yb, yab = f(a), f(b), f(2 * a - 3 * b)
    err = (yab - (2 * ya - 3 * yb)).abs().max() / yab.abs().max()
    assert err.item() < 1e-5
    assert torch.equal(f(a), ya)  # no atomics in the forward: run-to-run bit-identical
    # centre-only weights: the conv reduces to a per-row matmul
    wc = torch.zeros_like(w)
    wc[13] = w[13]
    yc = ops.conv_forward(a, ops.weight_image(wc), rb2.out_nbr, k, 27, 64, 64)
    assert ((yc - a @ w[13]).abs().max() / yc.abs().max()).item() < 1e-5


def test_dense_matches_oracle(oracle, cuda):
    rng = np.random.default_rng(3)
    shape = [2, 200, 176]
    cells = rng.choice(2 * 2 * 200 * 176, 15000, replace=False)
    coors = np.stack([cells // (2 * 200 * 176), (cells // (200 * 176)) % 2, (cells // 176) % 200,
                      cells % 176], 1).astype(np.int32)
    x = rng.standard_normal((15000, 64)).astype(np.float32)
    ref = oracle.dense(x, coors, 2, shape)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    out = ops.dense(xt, torch.from_numpy(coors).to(cuda), 2, shape)
    assert out.shape == (2, 64, 2, 200, 176)
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    gout = torch.randn_like(out)
    out.backward(gout)
    c = torch.from_numpy(coors).long().to(cuda)
    ref_g = gout[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    assert torch.equal(xt.grad, ref_g)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (32, 64), (20, 40), (128, 128)])
@pytest.mark.parametrize("sets,tile", [(1, 32), (2, 32), (1, 17), (2, 25), (2, 16)])
def test_tile_compacted_variants_match_oracle(oracle, cuda, cin, cout, sets, tile):
    """Every launch-time variant of the tile-compacted forward (one / two wave sets, tile heights
    16..32) against the oracle, pinned with lcf_set_option; two sets are run-to-run deterministic."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(cin + 7 * cout + sets + tile)
    coors = _frame_coors(oracle, 3, 7000)
    extra = coors.copy()
    extra[:, 2] = np.minimum(extra[:, 2] + 1, 1599)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 1, 1, 1, True)
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    L = _lib.lib()
    try:
        assert L.lcf_set_option(_lib.OPT_CONV_IMPL, 2) == 0
        assert L.lcf_set_option(_lib.OPT_CONV_SETS, sets) == 0
        assert L.lcf_set_option(_lib.OPT_CONV_TILE, tile) == 0
        out = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, rb.k_out, 27, cin, cout)
        again = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, rb.k_out, 27, cin, cout)
    finally:
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 0)
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)
        L.lcf_set_option(_lib.OPT_CONV_TILE, 0)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    assert torch.equal(out, again)


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (32, 64), (64, 32)])
@pytest.mark.parametrize("sets,tile,subm", [(0, 0, True), (1, 32, True), (2, 17, True),
                                            (3, 25, False), (4, 32, False), (4, 16, True)])
def test_quad_row_form_matches_oracle(oracle, cuda, cin, cout, sets, tile, subm):
    """cv_q4_kernel (csrc/conv_q4.h: v_mfma_f32_4x4x1 with the A-matrix broadcast, 4- or 8-row
    groups gathered straight into registers), every number of waves per tile and several tile
    heights, submanifold and strided, against the oracle; run-to-run deterministic; the same
    through a tile plan and with the row count on the device, bit for bit."""
    from objectdetection_amd import _lib
    from objectdetection_amd._lib import check, i3
    rng = np.random.default_rng(cin + 7 * cout + sets + tile)
    coors = _frame_coors(oracle, 4, 7000)
    extra = coors.copy()
    extra[:, 2] = np.minimum(extra[:, 2] + 1, 1599)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
        rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 1, 1, 1, True)
    else:
        out_coors, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = out_coors.shape[0]
        rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, False)
        assert rb.k_out == k_out
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    img = ops.weight_image(wt)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    try:
        assert L.lcf_set_option(_lib.OPT_CONV_IMPL, 5) == 0
        assert L.lcf_set_option(_lib.OPT_CONV_SETS, sets) == 0
        assert L.lcf_set_option(_lib.OPT_CONV_TILE, tile) == 0
        out = ops.conv_forward(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout)
        again = ops.conv_forward(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout)
        # a capacity with the count on the device, through a tile plan
        cap = k_out + 333
        tr = tile if tile else L.lcf_conv_tile_rows_for(k_out, 27, cin, cout, 0)
        nbr = torch.cat([rb.out_nbr, torch.full((333, 27), 5, dtype=torch.int32, device=cuda)])
        k_dev = torch.tensor([k_out], dtype=torch.int32, device=cuda)
        plan = torch.zeros(L.lcf_conv_tile_plan_ints(cap, 27, tr), dtype=torch.int32, device=cuda)
        check(L.lcf_conv_tile_plan(nbr.data_ptr(), cap, k_dev.data_ptr(), 27, tr,
                                   plan.data_ptr(), st))
        planned = torch.full((cap, cout), 7.0, device=cuda)
        ctx = _lib.Context()
        ctx.set(_lib.OPT_CONV_ROWS_HINT, k_out)
        check(L.lcf_indice_conv_fwd_planned(
            xt.data_ptr(), img.data_ptr(), 0, plan.data_ptr(), tr, cap, k_dev.data_ptr(), 27,
            cin, cout, None, None, 0, planned.data_ptr(), None, 1, i3(0), st, ctx.handle))
    finally:
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 0)
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)
        L.lcf_set_option(_lib.OPT_CONV_TILE, 0)
    assert rel_err(out.cpu().numpy(), ref) < RTOL
    assert torch.equal(out, again)
    assert torch.equal(planned[:k_out], out)
    assert bool((planned[k_out:] == 7.0).all())


@pytest.mark.parametrize("cin,cout,sets", [(64, 64, 1), (64, 64, 2), (16, 16, 1), (32, 128, 2),
                                           (128, 32, 1), (5, 64, 1)])
def test_conv_with_dense_epilogue_equals_conv_then_dense(oracle, cuda, cin, cout, sets):
    """lcf_indice_conv_fwd_dense (the stack's last SparseConv3d (3,1,1)/(2,1,1) + BN + ReLU with
    .dense() fused, middle.py:186-206) == lcf_indice_conv_fwd followed by lcf_dense, bit for bit,
    and == the oracle's conv + dense within the fp32 tolerance; batch of two scenes, a device-side
    row count below the capacity, and rows outside the dense shape dropped."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(cin * 3 + cout + sets)
    cs = []
    for b in range(2):
        c = _frame_coors(oracle, 11 + b, 9000)
        c[:, 1:] //= 8
        c[:, 0] = b
        cs.append(np.unique(c, axis=0))
    coors = np.concatenate(cs, 0).astype(np.int32)
    rng.shuffle(coors, axis=0)
    coors = np.ascontiguousarray(coors)
    shape = [5, 200, 176]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 1, 1, cin, cout)) / np.sqrt(3 * cin)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    oc, pairs, pn, oshape = oracle.rulebook_conv(coors, 2, shape, [3, 1, 1], [2, 1, 1], 0)
    assert list(oshape) == [2, 200, 176]
    ref = np.maximum(oracle.indice_conv_fwd(x, w, pairs, pn, oc.shape[0]) * sc[None] + sh[None], 0)
    ref_dense = oracle.dense(ref, oc, 2, oshape)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 2, shape, [3, 1, 1], [2, 1, 1], 0,
                            1, False)
    assert rb.k_out == oc.shape[0]
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    sct, sht = torch.from_numpy(sc).to(cuda), torch.from_numpy(sh).to(cuda)
    wimg = ops.weight_image(wt)
    L = _lib.lib()
    try:
        assert L.lcf_set_option(_lib.OPT_CONV_IMPL, 2) == 0
        assert L.lcf_set_option(_lib.OPT_CONV_SETS, sets) == 0
        rows = ops.conv_forward(xt, wimg, rb.out_nbr, rb.k_out, 3, cin, cout, sct, sht, True)
        two = ops.dense(rows, rb.out_indices, 2, oshape)
        one = ops.conv_forward_dense(xt, wimg, rb.out_nbr, rb.k_out, 3, cin, cout,
                                     rb.out_indices, 2, oshape, sct, sht, True)
        assert torch.equal(one, two)
        assert rel_err(one.cpu().numpy(), ref_dense) < RTOL
        # capacity > device-side count: rows past the count are not written
        cnt = rb.k_out - 37
        kd = torch.tensor([cnt], dtype=torch.int32, device=cuda)
        part = ops.conv_forward_dense(xt, wimg, rb.out_nbr, rb.k_out, 3, cin, cout,
                                      rb.out_indices, 2, oshape, sct, sht, True, n_rows_dev=kd)
        assert torch.equal(part, ops.dense(rows[:cnt], rb.out_indices[:cnt], 2, oshape))
        # a smaller dense shape: rows outside are dropped (as lcf_dense drops them)
        small = [2, 150, 100]
        crop = ops.conv_forward_dense(xt, wimg, rb.out_nbr, rb.k_out, 3, cin, cout,
                                      rb.out_indices, 1, small, sct, sht, True)
        assert torch.equal(crop, ops.dense(rows, rb.out_indices, 1, small))
        assert torch.equal(crop[0], two[0, :, :, :150, :100])
    finally:
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 0)
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)


X3_RTOL = 1e-4  # the north-star fp32 feature tolerance; measured: <= 1e-5


@pytest.mark.parametrize("cin,cout,subm,sets", [(32, 32, True, 1), (64, 64, True, 2),
                                                (32, 64, False, 1), (64, 128, True, 1),
                                                (128, 64, False, 2), (128, 128, True, 1)])
def test_bf16x3_forward_matches_oracle_at_fp32_tolerance(oracle, cuda, cin, cout, subm, sets):
    """Split-operand form (x = hi + lo in bf16, three matrix-core products, fp32 accumulation)
    against the oracle inside the fp32 tolerance, an order of magnitude tighter in practice; exact
    where hi + lo represents the operands exactly; the dense-target epilogue; determinism."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(cin * 5 + cout + sets)
    coors = _frame_coors(oracle, 4, 6000)
    extra = coors.copy()
    extra[:, 3] = np.minimum(extra[:, 3] + 1, 1407)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    # magnitudes over several octaves: the split must hold per element, not per tensor
    x = (rng.standard_normal((k, cin)) * np.exp2(rng.integers(-6, 7, (k, 1)))).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref = np.maximum(oracle.indice_conv_fwd(x, w, pairs, pn, k_out) * sc[None] + sh[None], 0)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    sct, sht = torch.from_numpy(sc).to(cuda), torch.from_numpy(sh).to(cuda)
    img = ops.weight_image_bf16x3(wt)
    L = _lib.lib()
    try:
        assert L.lcf_set_option(_lib.OPT_CONV_SETS, sets) == 0
        out = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout, sct, sht, True)
        again = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout, sct, sht,
                                        True)
        err = rel_err(out.cpu().numpy(), ref)
        assert err < X3_RTOL
        assert err < 2e-5  # what the construction gives (3 * 2^-18 per product, random signs)
        assert torch.equal(out, again)
        # operands with <= 16 significant bits are hi + lo exactly: what remains is the dropped
        # lo*lo term (<= 2^-18 per product) and the fp32 summation order
        def q16(a):
            return (a.view(np.uint32) & np.uint32(0xFFFFFF00)).view(np.float32)
        xq, wq = q16(x.copy()), q16(w.copy())
        xqt, wqt = torch.from_numpy(xq).to(cuda), torch.from_numpy(wq).to(cuda)
        a = ops.conv_forward_bf16x3(xqt, ops.weight_image_bf16x3(wqt), rb.out_nbr, rb.k_out, 27,
                                    cin, cout)
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 2)
        b = ops.conv_forward(xqt, ops.weight_image(wqt), rb.out_nbr, rb.k_out, 27, cin, cout)
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 4e-6
        # dense target
        if not subm:
            oshape = rb.out_shape
            sub = [int(v) for v in (oshape[0], 64, 64)]
            d = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout, sct, sht,
                                        True, dense_coors=rb.out_indices, batch_size=1, shape=sub)
            assert torch.equal(d, ops.dense(out, rb.out_indices, 1, sub))
    finally:
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 0)


@pytest.mark.parametrize("cin,cout,subm,sets", [(64, 64, True, 1), (32, 64, False, 2),
                                                (128, 128, True, 1), (64, 32, False, 1)])
def test_bf16x8_forward_is_fp32_grade(oracle, cuda, cin, cout, subm, sets):
    """Three-plane form (x = hi + mid + lo exactly, eight products): as close to the oracle as the
    fp32 MFMA kernel is (both ~1e-7 of the output's max magnitude), on rows spread over 13
    octaves; deterministic; the dense-target epilogue."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(cin * 9 + cout + sets)
    coors = _frame_coors(oracle, 8, 6000)
    extra = coors.copy()
    extra[:, 3] = np.minimum(extra[:, 3] + 1, 1407)
    coors = np.ascontiguousarray(np.unique(np.concatenate([coors, extra], 0), axis=0))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = (rng.standard_normal((k, cin)) * np.exp2(rng.integers(-6, 7, (k, 1)))).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    img = ops.weight_image_bf16x3(wt, planes=3)
    L = _lib.lib()
    try:
        assert L.lcf_set_option(_lib.OPT_CONV_SETS, sets) == 0
        out = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout, planes=3)
        again = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout, planes=3)
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 2)
        f32 = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, rb.k_out, 27, cin, cout)
        e8, e32 = rel_err(out.cpu().numpy(), ref), rel_err(f32.cpu().numpy(), ref)
        assert e8 < 1e-6 and e8 < 3 * e32 + 1e-7, (e8, e32)
        assert torch.equal(out, again)
        if not subm:
            sub = [int(rb.out_shape[0]), 64, 64]
            d = ops.conv_forward_bf16x3(xt, img, rb.out_nbr, rb.k_out, 27, cin, cout,
                                        dense_coors=rb.out_indices, batch_size=1, shape=sub,
                                        planes=3)
            assert torch.equal(d, ops.dense(out, rb.out_indices, 1, sub))
    finally:
        L.lcf_set_option(_lib.OPT_CONV_SETS, 0)
        L.lcf_set_option(_lib.OPT_CONV_IMPL, 0)


def test_bf16x3_identity_table_is_a_row_wise_linear_layer(cuda):
    """nbr = NULL, kvol = 1 (the row-wise linear form of the fp32 entry point) on split operands:
    against the float64 product, ragged row counts, device-side count below the capacity."""
    rng = np.random.default_rng(11)
    for n, cin, cout in ((1, 32, 32), (77, 64, 128), (5000, 128, 64)):
        x = rng.standard_normal((n, cin)).astype(np.float32)
        w = (rng.standard_normal((1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
        ref = x.astype(np.float64) @ w[0].astype(np.float64)
        xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
        img = ops.weight_image_bf16x3(wt)
        out = ops.conv_forward_bf16x3(xt, img, None, n, 1, cin, cout)
        assert rel_err(out.cpu().numpy(), ref) < 2e-5
        if n > 10:
            kd = torch.tensor([n - 9], dtype=torch.int32, device=cuda)
            part = torch.full((n, cout), 7.0, device=cuda)
            ops.conv_forward_bf16x3(xt, img, None, n, 1, cin, cout, n_rows_dev=kd, out=part)
            assert torch.equal(part[:n - 9], out[:n - 9])
            assert bool((part[n - 9:] == 7.0).all())


def test_bf16x3_rejects_unsupported_widths(cuda):
    w = torch.zeros((27, 16, 32), device=cuda)
    with pytest.raises(RuntimeError):
        ops.weight_image_bf16x3(w)


def test_dense_as_channels_last_bf16_equals_dense_view_cast(oracle, cuda):
    """lcf_dense_nhwc_bf16 == dense() -> view(N, C*D, H, W) -> channels_last -> bfloat16 of the
    reference's hand-off to the RPN (middle.py:206-209), bit for bit, forward and backward;
    also against the oracle's dense()."""
    rng = np.random.default_rng(5)
    cs = []
    for b in range(3):
        c = _frame_coors(oracle, 30 + b, 4000)
        c[:, 1] = c[:, 1] * 2 // 41
        c[:, 2:] //= 8
        c[:, 0] = b
        cs.append(np.unique(c, axis=0))
    coors = np.ascontiguousarray(np.concatenate(cs, 0).astype(np.int32))
    shape = [2, 200, 176]
    k, ch = coors.shape[0], 64
    x = rng.standard_normal((k, ch)).astype(np.float32)
    ref = oracle.dense(x, coors, 3, shape).reshape(3, ch * 2, 200, 176)
    ct = torch.from_numpy(coors).to(cuda)
    xa = torch.from_numpy(x).to(cuda).requires_grad_(True)
    xb = torch.from_numpy(x).to(cuda).requires_grad_(True)
    one = ops.dense_nhwc_bf16(xa, ct, 3, shape)
    assert one.dtype == torch.bfloat16 and one.shape == (3, ch * 2, 200, 176)
    assert one.is_contiguous(memory_format=torch.channels_last)
    two = ops.dense(xb, ct, 3, shape).view(3, ch * 2, 200, 176).contiguous(
        memory_format=torch.channels_last).to(torch.bfloat16)
    assert torch.equal(one, two)
    assert torch.equal(one.float().cpu(), torch.from_numpy(ref).to(torch.bfloat16).float())
    g = torch.randn(one.shape, device=cuda).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    one.backward(g)
    two.backward(g)
    assert torch.equal(xa.grad, xb.grad)
    # rows outside the shape are dropped, and get no gradient
    small = [2, 120, 90]
    xc = torch.from_numpy(x).to(cuda).requires_grad_(True)
    xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
    a = ops.dense_nhwc_bf16(xc, ct, 2, small)
    b = ops.dense(xd, ct, 2, small).view(2, ch * 2, 120, 90).to(torch.bfloat16)
    assert torch.equal(a, b)
    a.float().sum().backward()
    b.float().sum().backward()
    assert torch.equal(xc.grad, xd.grad)


def test_conv_with_dense_epilogue_rejects_what_it_cannot_do(cuda):
    from objectdetection_amd import _lib
    L = _lib.lib()
    t = torch.zeros(4096, device=cuda)
    i = torch.zeros(4096, dtype=torch.int32, device=cuda)
    a = lambda cout, nbr: L.lcf_indice_conv_fwd_dense(
        t.data_ptr(), t.data_ptr(), nbr, 8, None, 3, 16, cout, None, None, 0, i.data_ptr(), 1,
        (ctypes.c_int * 3)(1, 2, 4), t.data_ptr(), None)
    assert a(24, i.data_ptr()) == -3  # LCF_ERR_UNSUPPORTED: padded output width
    assert a(16, None) == -1  # LCF_ERR_ARG: no neighbour table
    torch.cuda.synchronize()


def test_option_values_are_validated(cuda):
    from objectdetection_amd import _lib
    L = _lib.lib()
    assert L.lcf_set_option(_lib.OPT_CONV_TILE, 15) != 0
    assert L.lcf_set_option(_lib.OPT_CONV_TILE, 33) != 0
    assert L.lcf_set_option(_lib.OPT_CONV_SETS, 5) != 0
    assert L.lcf_set_option(_lib.OPT_CONV_IMPL, 6) != 0
    assert L.lcf_set_option(99, 0) != 0
    # options that change what a call assumes about its buffers exist per context only
    assert L.lcf_set_option(_lib.OPT_WS_CLEAN, 1) != 0
    assert L.lcf_set_option(_lib.OPT_NBR_CLEAN, 1) != 0
    assert L.lcf_set_option(_lib.OPT_CONV_ROWS_HINT, 100) != 0
    ctx = _lib.Context()
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_WS_CLEAN, 2) != 0
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_NBR_CLEAN, 2) != 0
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_NBR_CLEAN, 1) == 0
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_CONV_ROWS_HINT, -1) != 0
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_CONV_TILE, 15) != 0
    assert L.lcf_ctx_set_option(ctx.handle, _lib.OPT_CONV_TILE, 24) == 0
    assert L.lcf_ctx_set_option(None, _lib.OPT_CONV_TILE, 24) != 0
    assert L.lcf_ctx_set_option(ctx.handle, 99, 0) != 0


@pytest.mark.parametrize("cin,cout,subm", [(32, 32, True), (64, 64, True), (32, 64, False),
                                           (64, 128, True), (128, 64, False)])
def test_bf16_forward_and_dgrad_match_oracle(oracle, cuda, cin, cout, subm):
    """bf16-operand MFMA variant (training) vs the fp32 oracle, within the stated tolerance, and
    exactly equal to the fp32 kernel on inputs that bf16 represents exactly."""
    rng = np.random.default_rng(cin + cout)
    coors = _frame_coors(oracle, 2, 6000)
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, 1, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, 1, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), 1, shape, 3, 2, 1, 1, subm)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    wt = torch.from_numpy(w).to(cuda).requires_grad_(True)
    with ops.conv_math("bf16"):
        out = ops.indice_conv(xt, wt, rb)
        g = rng.standard_normal((k_out, cout)).astype(np.float32)
        out.backward(torch.from_numpy(g).to(cuda))
    assert rel_err(out.detach().cpu().numpy(), ref) < BF16_RTOL
    din_ref, dw_ref = oracle.indice_conv_bwd(x, w, g, pairs, pn)
    assert rel_err(xt.grad.cpu().numpy(), din_ref) < BF16_RTOL
    assert rel_err(wt.grad.cpu().numpy(), dw_ref) < RTOL  # wgrad stays fp32
    # bf16-exact inputs: products are exact in fp32, only the summation order differs
    xe = torch.from_numpy(x).to(cuda).bfloat16().float()
    we = torch.from_numpy(w).to(cuda).bfloat16().float()
    a = ops.conv_forward_bf16(xe, ops.weight_image_bf16(we), rb.out_nbr, rb.k_out, 27, cin, cout)
    b = ops.conv_forward(xe, ops.weight_image(we), rb.out_nbr, rb.k_out, 27, cin, cout)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("k,c", [(1, 16), (37, 16), (5000, 32), (150001, 64), (3000, 128),
                                 (777, 12)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_relu_training_kernels_match_torch(cuda, k, c, relu):
    """lcf_bn_relu_fwd / _bwd (training BatchNorm1d + ReLU between the sparse convs,
    middle.py:124-125) against torch's BatchNorm1d + ReLU in fp32: output, input / weight / bias
    gradients, running statistics; run-to-run bit-identical."""
    import copy
    torch.manual_seed(k + c)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(cuda).train()
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, std=0.3)
    ref = copy.deepcopy(bn)
    x = (torch.randn(k, c, device=cuda) * 2 + 0.7)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(k, c, device=cuda)
    ya = ops.bn_relu_train(bn, xa, relu)
    if k > 1:
        yb = ref(xb)
        yb = torch.relu(yb) if relu else yb
    else:  # torch refuses a batch of one in training; the kernels give the k = 1 limit (x - x = 0)
        yb = torch.relu(ref.bias.expand(1, c)) if relu else ref.bias.expand(1, c).clone()
    tol = 2e-5
    scale = max(float(yb.detach().abs().max()), 1e-6)
    assert float((ya - yb).detach().abs().max()) / scale < tol
    if k > 1:
        (ya * g).sum().backward()
        (yb * g).sum().backward()
        for a, b, name in [(xa.grad, xb.grad, "dx"), (bn.weight.grad, ref.weight.grad, "dgamma"),
                           (bn.bias.grad, ref.bias.grad, "dbeta")]:
            assert float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6) < 1e-4, name
        assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
    again = ops.bn_relu_train(copy.deepcopy(ref), xa.detach(), relu)
    again2 = ops.bn_relu_train(copy.deepcopy(ref), xa.detach(), relu)
    assert torch.equal(again, again2)


@pytest.mark.parametrize("math", [0, 2, 3])
def test_planned_forward_is_bit_identical_to_the_table_forward(cuda, math):
    """lcf_conv_tile_plan + lcf_indice_conv_fwd_planned: the per-tile set-up computed once per
    rulebook gives exactly what the per-launch set-up gives (same kernels, same order)."""
    from objectdetection_amd import _lib, ops, synth
    from objectdetection_amd._lib import check, i3
    L = _lib.lib()
    pts = torch.from_numpy(synth.vkitti_frame(9, 12000)).to(cuda)
    _, coors, _, vnum = ops.voxelize(pts, synth.VOXEL_SIZE, synth.PC_RANGE, 5, 40000, batch_idx=0)
    coors = coors[:int(vnum.item())].contiguous()
    rb = ops.build_rulebook(coors, 1, [41, 1600, 1408], 3, 1, 0, 1, True)
    k, kvol = rb.k_out, rb.kvol
    g = torch.Generator(device=cuda).manual_seed(3)
    for cin, cout in ((32, 64), (64, 64)):
        x = torch.randn((k, cin), device=cuda, generator=g)
        w = torch.randn((3, 3, 3, cin, cout), device=cuda, generator=g) * 0.1
        scale = torch.rand(cout, device=cuda, generator=g) + 0.5
        shift = torch.randn(cout, device=cuda, generator=g)
        if math == 0:
            img = ops.weight_image(w)
            ref = ops.conv_forward(x, img, rb.out_nbr, k, kvol, cin, cout, scale, shift, True)
        else:
            img = ops.weight_image_bf16x3(w, planes=math)
            ref = ops.conv_forward_bf16x3(x, img, rb.out_nbr, k, kvol, cin, cout, scale, shift,
                                          True, planes=math)
        st = torch.cuda.current_stream().cuda_stream
        for cap_extra in (0, 777):  # exact row count, and a capacity with the count on the device
            cap = k + cap_extra
            # (the height the table forward of that shape and arithmetic cuts)
            tr = L.lcf_conv_tile_rows_for(k, kvol, cin, cout, math)
            nbr = torch.cat([rb.out_nbr, torch.full((cap_extra, kvol), 5, dtype=torch.int32,
                                                    device=cuda)])
            k_dev = torch.tensor([k], dtype=torch.int32, device=cuda)
            plan = torch.zeros(L.lcf_conv_tile_plan_ints(cap, kvol, tr), dtype=torch.int32,
                               device=cuda)
            check(L.lcf_conv_tile_plan(nbr.data_ptr(), cap, k_dev.data_ptr(), kvol, tr,
                                       plan.data_ptr(), st))
            out = torch.full((cap, cout), 7.0, device=cuda)
            check(L.lcf_indice_conv_fwd_planned(
                x.data_ptr(), img.data_ptr(), math, plan.data_ptr(), tr, cap, k_dev.data_ptr(),
                kvol, cin, cout, scale.data_ptr(), shift.data_ptr(), 1, out.data_ptr(), None, 1,
                i3(0), st))
            # the table forward picks one or two wave sets from the rows it expects: pin both
            # runs to the same choice through the rows the capacity implies
            ctx = _lib.Context()
            ctx.set(_lib.OPT_CONV_ROWS_HINT, cap)
            assert torch.equal(out[:k], ref) or cap_extra, (cin, cout)
            if cap_extra:
                assert (out[:k] - ref).abs().max() <= 1e-5 * ref.abs().max()
            assert bool((out[k:] == 7.0).all())


@pytest.mark.parametrize("cin,cout,subm,batch", [(32, 32, True, 1), (64, 64, True, 2),
                                                 (32, 64, False, 1), (64, 64, False, 2)])
def test_bf16_row_kernels_match_oracle(oracle, cuda, cin, cout, subm, batch):
    """The training kernels on rows STORED as bf16 (lcf_indice_conv_fwd_b16 / _wgrad_b16, called
    through the C ABI): forward, dgrad (in_nbr + transposed image, bf16 output) and wgrad against
    the oracle run on the same bf16-rounded operands -- products are then exact in fp32, only the
    summation order and the final rounding differ -- and within the stated bf16 tolerance of the
    oracle on the unrounded operands."""
    from objectdetection_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(100 + cin + cout + batch)
    parts = []
    for b in range(batch):
        c = _frame_coors(oracle, 2 + b, 5000)
        c[:, 0] = b
        parts.append(c)
    coors = np.ascontiguousarray(np.concatenate(parts))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = rng.standard_normal((k, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, batch, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, batch, shape, 3, 2, 1)
        k_out = oc.shape[0]
    g = rng.standard_normal((k_out, cout)).astype(np.float32)
    rnd = lambda a: torch.from_numpy(a).bfloat16().float().numpy()
    xr, wr, gr = rnd(x), rnd(w), rnd(g)
    ref = oracle.indice_conv_fwd(xr, wr, pairs, pn, k_out)
    din_ref, dw_ref = oracle.indice_conv_bwd(xr, wr, gr, pairs, pn)
    ref_full = oracle.indice_conv_fwd(x, w, pairs, pn, k_out)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), batch, shape, 3, 2, 1, 1, subm)
    assert rb.k_out == k_out
    st = torch.cuda.current_stream().cuda_stream
    xb = torch.from_numpy(x).to(cuda).bfloat16().contiguous()
    gb = torch.from_numpy(g).to(cuda).bfloat16().contiguous()
    wt = torch.from_numpy(w).to(cuda)
    wimg, wimg_t = ops.weight_image_bf16(wt), ops.weight_image_bf16(wt, transpose=True)
    out = torch.empty((k_out, cout), device=cuda)
    assert L.lcf_indice_conv_fwd_b16(xb.data_ptr(), k, wimg.data_ptr(), rb.out_nbr.data_ptr(),
                                     k_out, 27, cin, cout, out.data_ptr(), 0, st) == 0
    din = torch.empty((k, cin), dtype=torch.bfloat16, device=cuda)
    assert L.lcf_indice_conv_fwd_b16(gb.data_ptr(), k_out, wimg_t.data_ptr(),
                                     rb.in_nbr.data_ptr(), k, 27, cout, cin, din.data_ptr(), 1,
                                     st) == 0
    dw = torch.empty((27, cin, cout), device=cuda)
    assert L.lcf_indice_conv_wgrad_b16(xb.data_ptr(), gb.data_ptr(), rb.in_nbr.data_ptr(), k, 27,
                                       cin, cout, dw.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref) < 1e-5            # fp32 sums of exact products
    assert rel_err(out.cpu().numpy(), ref_full) < BF16_RTOL
    assert rel_err(din.float().cpu().numpy(), din_ref) < 4e-3  # + one bf16 rounding of the result
    assert rel_err(dw.cpu().numpy().reshape(dw_ref.shape), dw_ref) < 1e-5
    # every G (rows per wave) cuts the same sums: bit-identical outputs
    out2 = ops.conv_forward_b16(xb, wimg, rb.out_nbr, k_out, 27, cin, cout)
    assert torch.equal(out, out2)
    # unsupported shapes and missing operands are refused
    assert L.lcf_indice_conv_fwd_b16(xb.data_ptr(), k, wimg.data_ptr(), rb.out_nbr.data_ptr(),
                                     k_out, 27, 16, cout, out.data_ptr(), 0, st) == -3
    assert L.lcf_indice_conv_fwd_b16(xb.data_ptr(), k, wimg.data_ptr(), None, k_out, 27, cin, cout,
                                     out.data_ptr(), 0, st) == -1
    assert L.lcf_indice_conv_wgrad_b16(xb.data_ptr(), gb.data_ptr(), rb.in_nbr.data_ptr(), k, 27,
                                       128, 128, dw.data_ptr(), st) == -3


@pytest.mark.parametrize("k,c", [(37, 32), (150001, 64), (3000, 128)])
def test_bn_relu_bf16_rows_and_in_kernel_running_stats(cuda, k, c):
    """lcf_bn_relu_fwd_ex / _bwd_ex: y stored as bf16 equals the fp32 kernel's y rounded to
    nearest even; with bf16 dy in and bf16 dx out the gradients equal the fp32 kernels' on the
    same (rounded) operands; the forward's own running-statistics update equals nn.BatchNorm1d's."""
    import copy
    torch.manual_seed(k + c)
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(cuda).train()
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.normal_(bn.bias, std=0.3)
    ref = copy.deepcopy(bn)
    x = (torch.randn(k, c, device=cuda) * 2 + 0.7)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = ops.bn_relu_train(bn, xa, True, y_bf16=True)
    yb = ops.bn_relu_train(ref, xb, True)
    assert ya.dtype == torch.bfloat16 and torch.equal(ya, yb.bfloat16())
    g = torch.randn(k, c, device=cuda).bfloat16()
    ya.backward(g)
    yb.backward(g.float())
    # (the ReLU mask comes from y: bf16 y > 0 exactly where fp32 y > 0)
    scale = float(xb.grad.abs().max())
    assert float((xa.grad - xb.grad).abs().max()) / scale < 1e-6
    assert torch.allclose(bn.weight.grad, ref.weight.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(bn.bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-5)
    # running statistics: updated inside the forward kernel, against torch's own module
    tbn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(cuda).train()
    tbn.load_state_dict({k_: v for k_, v in ref.state_dict().items()
                         if "running" not in k_ and "num_batches" not in k_}, strict=False)
    tbn(x)
    assert int(bn.num_batches_tracked) == 1 == int(tbn.num_batches_tracked)
    assert torch.allclose(bn.running_mean, tbn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, tbn.running_var, rtol=1e-4, atol=1e-6)


def test_all_ones_kernel_is_a_row_gemm_at_any_width(cuda):
    """spconv's 1x1x1 short cut (what second/pytorch/models/resnet.py:20-29 and SparseBottleneck's
    64 -> 256 expansion, resnet.py:69-110, rely on): no rulebook, a row-wise GEMM with autograd,
    channel counts beyond the sparse kernels' 128 included; indices pass through unchanged."""
    from objectdetection_amd import spconv
    torch.manual_seed(0)
    k = 5000
    idx = torch.stack([torch.zeros(k), torch.randint(0, 41, (k,)), torch.randint(0, 1600, (k,)),
                       torch.randint(0, 1408, (k,))], 1).int().to(cuda)
    for cin, cout, bias in ((64, 256, False), (256, 64, True), (16, 16, True)):
        conv = spconv.SubMConv3d(cin, cout, 1, bias=bias, indice_key="b").to(cuda)
        x = torch.randn(k, cin, device=cuda, requires_grad=True)
        y = conv(spconv.SparseConvTensor(x, idx, [41, 1600, 1408], 1))
        assert y.indices is idx and not y.indice_dict
        ref = x @ conv.weight.view(cin, cout) + (conv.bias if bias else 0)
        assert torch.allclose(y.features, ref, rtol=1e-5, atol=1e-5)
        y.features.sum().backward()
        assert conv.weight.grad is not None and x.grad.shape == x.shape
        with torch.no_grad():  # the inference path with a folded BatchNorm + ReLU epilogue
            sc, sh = torch.rand(cout, device=cuda) + 0.5, torch.randn(cout, device=cuda)
            z = conv(spconv.SparseConvTensor(x.detach(), idx, [41, 1600, 1408], 1), _scale=sc,
                     _shift=sh, _relu=True)
            assert torch.allclose(z.features, torch.relu(ref * sc + sh), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,cout,subm,batch", [(32, 32, True, 1), (64, 64, True, 2),
                                                 (32, 64, False, 1), (64, 64, False, 1)])
def test_six_product_planes_form_is_fp32_grade(oracle, cuda, cin, cout, subm, batch):
    """lcf_indice_conv_fwd_x6 (three exact bf16 planes, six products; csrc/conv_bx.h) through the C
    ABI: within the fp32 tolerance of the oracle (1e-4), and at the SAME distance from a float64
    reference as the fp32 MFMA kernel on the same inputs (both a few 1e-7 of the result's max
    magnitude; the bound asserted is 2.5 x the fp32 kernel's own distance + 1e-7).  Planes written by
    the kernel's epilogue sum to its fp32 output exactly; a second conv fed with those planes
    matches the fp32 kernel fed with the fp32 rows; a device-side count below the capacity leaves
    the rows beyond it untouched."""
    from objectdetection_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7 + cin + cout + batch)
    parts = []
    for b in range(batch):
        c = _frame_coors(oracle, 4 + b, 6000)
        c[:, 0] = b
        parts.append(c)
    coors = np.ascontiguousarray(np.concatenate(parts))
    shape = [41, 1600, 1408]
    k = coors.shape[0]
    x = np.abs(rng.standard_normal((k, cin))).astype(np.float32) * rng.random((k, 1), np.float32)
    w = (rng.standard_normal((3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    scale = (rng.random(cout) + 0.5).astype(np.float32)
    shift = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    if subm:
        pairs, pn = oracle.rulebook_subm(coors, batch, shape, 3)
        k_out = k
    else:
        oc, pairs, pn, _ = oracle.rulebook_conv(coors, batch, shape, 3, 2, 1)
        k_out = oc.shape[0]
    ref32 = np.maximum(oracle.indice_conv_fwd(x, w, pairs, pn, k_out) * scale + shift, 0)
    rb = ops.build_rulebook(torch.from_numpy(coors).to(cuda), batch, shape, 3, 2, 1, 1, subm)
    assert rb.k_out == k_out
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    sc, sh = torch.from_numpy(scale).to(cuda), torch.from_numpy(shift).to(cuda)
    # float64 reference on the device
    wd = wt.double().reshape(27, cin, cout)
    ref64 = torch.zeros((k_out, cout), dtype=torch.float64, device=cuda)
    for o in range(27):
        idx = rb.out_nbr[:, o].long()
        v = idx >= 0
        ref64[v] += xt[idx[v]].double() @ wd[o]
    ref64 = torch.relu(ref64 * sc.double() + sh.double())
    dist = lambda t: float((t.double() - ref64).abs().max() / ref64.abs().max())
    y32 = ops.conv_forward(xt, ops.weight_image(wt), rb.out_nbr, k_out, 27, cin, cout, sc, sh, True)
    st = torch.cuda.current_stream().cuda_stream
    planes = torch.zeros((3, k + 5, cin), dtype=torch.bfloat16, device=cuda)
    assert L.lcf_split_planes(xt.data_ptr(), k, None, cin, planes.data_ptr(), k + 5, st) == 0
    assert torch.equal(planes[:, :k].float().sum(0), xt)  # h + m + l == x, exactly
    wimg3 = ops.weight_image_bf16x3(wt, planes=3)
    y = torch.full((k_out + 3, cout), 7.0, device=cuda)
    kd = torch.tensor([k_out], dtype=torch.int32, device=cuda)
    assert L.lcf_indice_conv_fwd_x6(planes.data_ptr(), k + 5, wimg3.data_ptr(),
                                    rb.out_nbr.data_ptr(), k_out + 3, kd.data_ptr(), 27, cin, cout,
                                    sc.data_ptr(), sh.data_ptr(), 1, y.data_ptr(), 0, 0, st) == 0
    yp = torch.zeros((3, k_out + 3, cout), dtype=torch.bfloat16, device=cuda)
    assert L.lcf_indice_conv_fwd_x6(planes.data_ptr(), k + 5, wimg3.data_ptr(),
                                    rb.out_nbr.data_ptr(), k_out + 3, kd.data_ptr(), 27, cin, cout,
                                    sc.data_ptr(), sh.data_ptr(), 1, yp.data_ptr(), 1, k_out + 3,
                                    st) == 0
    torch.cuda.synchronize()
    assert bool((y[k_out:] == 7.0).all()) and bool((yp[:, k_out:] == 0).all())
    assert rel_err(y[:k_out].cpu().numpy(), ref32) < 1e-4          # the oracle, fp32 tolerance
    d6, d32 = dist(y[:k_out]), dist(y32)
    assert d6 < 2.5 * d32 + 1e-7 and d6 < 1e-6, (d6, d32)           # fp32-grade
    assert torch.equal(yp[:, :k_out].float().sum(0), y[:k_out])     # the planes ARE the fp32 result
    y2 = ops.conv_forward_x6(planes, wimg3, rb.out_nbr, k_out, 27, cin, cout, sc, sh, True)
    assert torch.equal(y2, y[:k_out])                               # deterministic
    if subm and cin == cout:  # a second layer on the planes the first one wrote
        z6 = ops.conv_forward_x6(yp[:, :k_out].contiguous(), wimg3, rb.out_nbr, k_out, 27, cin, cout)
        z32 = ops.conv_forward(y32, ops.weight_image(wt), rb.out_nbr, k_out, 27, cin, cout)
        assert float((z6 - z32).abs().max() / z32.abs().max()) < 2e-6
    # refused: unsupported widths, planes that are not 16-byte multiples apart, missing table
    assert L.lcf_indice_conv_fwd_x6(planes.data_ptr(), k + 5, wimg3.data_ptr(),
                                    rb.out_nbr.data_ptr(), k_out, None, 27, 16, cout, None, None, 0,
                                    y.data_ptr(), 0, 0, st) == -3
    assert L.lcf_indice_conv_fwd_x6(planes.data_ptr(), k + 5, wimg3.data_ptr(), None, k_out, None,
                                    27, cin, cout, None, None, 0, y.data_ptr(), 0, 0, st) == -1
