#!/usr/bin/env python3
"""Diagnostic: how much of a frame is the feature branch alone?  Replays ONE cloud through (a) the
normal two-branch graph and (b) a graph that holds only the feature branch -- voxelizer, fused
VFE, the 14 convs with dense() in the last epilogue, the dense clear and the workspace re-arm --
reusing the rulebooks the previous full frame of the same cloud left in the engine's buffers.
(b) - the sum of its kernels = launch gaps; (a) - (b) = what sharing the chip with the rulebook
branch costs.  Results of (b) are only valid for the repeated cloud; never a product path.
Usage: python tools/diag_feature_branch.py [conv_math]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib, synth  # noqa: E402
from objectdetection_amd._lib import check, f3, f6, i3  # noqa: E402
from objectdetection_amd.engine import FrameEngine  # noqa: E402
from objectdetection_amd.frontend import SparseFrontEnd, randomize_  # noqa: E402


class FeatureOnly(FrameEngine):
    def _enqueue_clean(self):
        if not getattr(self, "feature_only", False):
            return super()._enqueue_clean()
        L = self.L
        cx = self.ctx.handle
        st = torch.cuda.current_stream(self.dev).cuda_stream
        P = lambda t: None if t is None else t.data_ptr()
        MV, T, F = self.max_voxels, self.T, self.F
        check(L.lcf_voxelize(P(self.points), self.max_points, P(self.n_dev), F, f3(self.vs),
                             f6(self.rng), T, MV, self.cap_mode, 0, 4, P(self.voxels),
                             P(self.coors), P(self.num), P(self.vnum), P(self.vox_ws),
                             self.vox_ws.numel(), st, cx))
        kd = self.vnum
        check(L.lcf_fill_u32(P(self.dense), 0, self.dense.numel() * 4, st))
        (w1, c0, u1, s1, h1), (w2, _, u2, s2, h2), (w3, _, u3, s3, h3) = self.lin
        check(L.lcf_vfe_fused(P(self.voxels), P(self.num), MV, P(kd), T, F, u1, u2, u3,
                              P(w1), P(s1), P(h1), P(w2), P(s2), P(h2), P(w3), P(s3), P(h3),
                              P(self.feat0), st))
        feat, coors, lvl = self.feat0, self.coors, 0
        for l in self.layers:
            o = l.rb_owner
            if l.subm:
                k_cap, k_dev = l.k_in_cap, kd
            else:
                k_cap, k_dev = l.k_out_cap, l.k_out
                lvl += 1
                coors, kd = l.out_coors, l.k_out
            self.ctx.set(_lib.OPT_CONV_ROWS_HINT, int(self.level_need[lvl]))
            last = l is self.layers[-1]
            if getattr(l, "planned", False):  # the plans of the previous full frame are resident
                check(L.lcf_indice_conv_fwd_planned(
                    P(feat), P(l.wimg), (3 if l.planes == 3 else 2) if l.x3 else 0,
                    P(self.chain.plan[o.level]), self.chain.plan_rows[o.level], k_cap, P(k_dev),
                    l.kvol, l.cin, l.cout, P(l.scale), P(l.shift), int(l.relu),
                    P(self.dense if last else l.out), P(coors) if last else None, 1,
                    i3(self.out_shape), st, cx))
            elif l.x3:
                fn = L.lcf_indice_conv_fwd_bf16x8 if l.planes == 3 else L.lcf_indice_conv_fwd_bf16x3
                check(fn(P(feat), P(l.wimg), P(o.nbr), k_cap, P(k_dev), l.kvol, l.cin, l.cout,
                         P(l.scale), P(l.shift), int(l.relu), P(self.dense if last else l.out),
                         P(coors) if last else None, 1, i3(self.out_shape), st, cx))
            elif last:
                check(L.lcf_indice_conv_fwd_dense(P(feat), P(l.wimg), P(o.nbr), k_cap, P(k_dev),
                                                  l.kvol, l.cin, l.cout, P(l.scale), P(l.shift),
                                                  int(l.relu), P(coors), 1, i3(self.out_shape),
                                                  P(self.dense), st, cx))
            else:
                check(L.lcf_indice_conv_fwd(P(feat), P(l.wimg), P(o.nbr), k_cap, P(k_dev), l.kvol,
                                            l.cin, l.cout, P(l.scale), P(l.shift), int(l.relu),
                                            P(l.out), st, cx))
            feat = l.out
        check(L.lcf_voxelize_workspace_reset(self.vox_ws.data_ptr(), self.vox_ws.numel(),
                                             self.max_points, self.T, st))


def rate(eng, c, n=400):
    for _ in range(40):
        eng.forward(c)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        eng.forward(c)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    cm = sys.argv[1] if len(sys.argv) > 1 else "f32"
    dev = torch.device("cuda:0")
    model = randomize_(SparseFrontEnd("M2").eval(), 0).to(dev)
    c = torch.from_numpy(synth.vkitti_frame(0, 20000)).to(dev)
    with torch.no_grad():
        eng = FeatureOnly(model, calib=[c], conv_math=cm)
        ref = eng.forward(c).clone()
        full = rate(eng, c)
        eng.feature_only = True
        eng._capture()
        out = eng.forward(c).clone()
        assert torch.equal(out, ref), "feature-only replay differs"
        feat = rate(eng, c)
        # (c) both branches, but the feature branch does not WAIT for the rulebooks (it reads the
        # identical ones the previous frame left): what remains above (b) is contention, what
        # (a) has above (c) is waiting on the rulebook events
        eng.feature_only = False
        skip = {id(e) for e in eng.rb_events}
        orig = torch.cuda.Stream.wait_event

        def wait_event(self, ev):
            if id(ev) not in skip:
                orig(self, ev)
        torch.cuda.Stream.wait_event = wait_event
        try:
            eng._capture()
        finally:
            torch.cuda.Stream.wait_event = orig
        out = eng.forward(c).clone()
        assert torch.equal(out, ref), "no-wait replay differs"
        nowait = rate(eng, c)
    print(f"{cm}: two-branch frame {full:.1f} us, feature branch alone {feat:.1f} us, "
          f"difference {full - feat:.1f} us; both branches without rulebook waits {nowait:.1f} us "
          f"(waiting {full - nowait:.1f} us, contention {nowait - feat:.1f} us)")


if __name__ == "__main__":
    main()
