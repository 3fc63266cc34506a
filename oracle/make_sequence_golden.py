"""Generate tests/golden/sequence_branch.npz from the reference's own "sequence"-branch modules (build container only).

TEST INFRASTRUCTURE.  Run:  python -m oracle.make_sequence_golden
ResNetLike, MotionDecoderV2 (out_dim 3 and 1), TransDSSL and transformation_from_parameters are imported from /root/reference through
`oracle/ref_loader.load_sequence`, filled with the name-hashed weights of `oracle/fill.py` (BatchNorm running statistics included),
run in eval mode on seeded Swin-T-shaped feature maps of a 64 x 96 frame pair, and composed exactly as OneFormer.forward composes them
(model/oneformer_model.py:306-365).  Inputs + outputs are stored; weights and source are not.
"""
import os
import warnings

import numpy as np
import torch

from . import fill, ref_loader
from . import sequence_ref as S

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    warnings.filterwarnings("ignore")
    assert ref_loader.available(), "needs /root/reference"
    ref = ref_loader.load_sequence()
    g = torch.Generator().manual_seed(77)
    B, H, W = 1, 64, 96
    cur, prev = torch.randn(B, 3, H, W, generator=g), torch.randn(B, 3, H, W, generator=g)
    fc = {f"res{i + 2}": torch.randn(B, c, H // s, W // s, generator=g) for i, (c, s) in enumerate(zip(S.SWIN_T_CH, (4, 8, 16, 32)))}
    fp = {k: torch.randn(v.shape, generator=g) for k, v in fc.items()}
    pose = ref.pose.ResNetLike(); fill.fill_module(pose, "pose_decoder."); pose.eval()
    flow = ref.motion.MotionDecoderV2(num_input_images=2, out_dim=3); fill.fill_module(flow, "motion_decoder."); flow.eval()
    mask = ref.motion.MotionDecoderV2(num_input_images=2, out_dim=1); fill.fill_module(mask, "motion_mask."); mask.eval()
    depth = ref.transdssl.TransDSSL(None, None); fill.fill_module(depth, "sem_seg_head.depth_decoder."); depth.eval()
    # the state-dict names / shapes the restatement assumes are the reference's
    got = {}
    for pfx, m in (("pose_decoder.", pose), ("motion_decoder.", flow), ("motion_mask.", mask), ("sem_seg_head.depth_decoder.", depth)):
        got.update({pfx + k: tuple(v.shape) for k, v in m.state_dict().items()})
    assert got == S.sequence_param_shapes(), set(got) ^ set(S.sequence_param_shapes())
    with torch.no_grad():
        fm = {k: torch.cat([fp[k], fc[k]], 1) for k in fc}                                       # oneformer_model.py:318-322
        axis, trans = pose(fm)
        axis, trans = axis[:, 0], trans[:, 0]
        cam = ref.geometry.transformation_from_parameters(axis, trans, invert=True)
        cam_fwd = ref.geometry.transformation_from_parameters(axis, trans, invert=False)
        mo = {"motion_input": {"full_res_input": torch.cat([prev, cur], 1), **fm}}
        ego = torch.cat((trans, axis), -1).permute(0, 2, 1).unsqueeze(3)
        f_out, m_out = flow(mo, ego), mask(mo, ego)
        d_out = depth.forward_features(fc)
    arrs = {"cur": cur, "prev": prev, **{"cur_" + k: v for k, v in fc.items()}, **{"prev_" + k: v for k, v in fp.items()},
            "axisangle": axis, "translation": trans, "cam_T_cam": cam, "cam_T_cam_not_inverted": cam_fwd,
            **{f"flow{s}": f_out[("complete_flow", s)] for s in range(4)}, **{f"motion_mask{s}": m_out[("motion_mask", s)] for s in range(4)},
            "motion_prob0": m_out[("motion_prob", 0)], **{f"disp{s}": d_out[("disp", s)] for s in range(4)}}
    path = os.path.join(OUT, "sequence_branch.npz")
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in arrs.items()})
    print(f"sequence_branch: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
