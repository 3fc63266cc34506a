"""The oracle's "sequence"-branch restatement (oracle/sequence_ref.py) against the fixture the reference's own modules produced
(tests/golden/sequence_branch.npz, oracle/make_sequence_golden.py).  CPU only."""
import torch

from conftest import load_golden
from oracle import fill
from oracle import sequence_ref as S


def _inputs(g):
    fc = {f"res{i}": g[f"cur_res{i}"] for i in range(2, 6)}
    fp = {f"res{i}": g[f"prev_res{i}"] for i in range(2, 6)}
    return fc, fp


def test_sequence_branch_restatement_matches_the_reference():
    g = load_golden("sequence_branch")
    sd = fill.state_dict_for(S.sequence_param_shapes())
    fc, fp = _inputs(g)
    with torch.no_grad():
        out = S.sequence_forward(g["cur"], g["prev"], fc, fp, sd)
        fm = {k: torch.cat([fp[k], fc[k]], 1) for k in fc}
        ego = torch.cat((out["translation"], out["axisangle"]), -1).permute(0, 2, 1).unsqueeze(3)
        mi = {"full_res_input": torch.cat([g["prev"], g["cur"]], 1), **fm}
        flow = S.motion_decoder_v2(mi, ego, sd, "motion_decoder", 3)
        mask = S.motion_decoder_v2(mi, ego, sd, "motion_mask", 1)
        disp = S.transdssl(fc, sd)
    tol = dict(atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(out["axisangle"], g["axisangle"], **tol)
    torch.testing.assert_close(out["translation"], g["translation"], **tol)
    torch.testing.assert_close(out["cam_T_cam"], g["cam_T_cam"], **tol)
    torch.testing.assert_close(S.transformation_from_parameters(g["axisangle"], g["translation"]), g["cam_T_cam_not_inverted"], **tol)
    for s in range(4):
        torch.testing.assert_close(flow[("complete_flow", s)], g[f"flow{s}"], **tol)
        torch.testing.assert_close(mask[("motion_mask", s)], g[f"motion_mask{s}"], **tol)
        torch.testing.assert_close(disp[("disp", s)], g[f"disp{s}"], **tol)
    torch.testing.assert_close(mask[("motion_prob", 0)], g["motion_prob0"], **tol)
    torch.testing.assert_close(out["disp_results"], g["disp0"], **tol)
    # pose algebra: the inverted transform is the inverse of the forward one
    eye = out["cam_T_cam"] @ g["cam_T_cam_not_inverted"]
    torch.testing.assert_close(eye, torch.eye(4).expand_as(eye), atol=1e-5, rtol=0)
