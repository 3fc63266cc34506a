"""CPU restatement (fp32, plain PyTorch, functional) of the reference's "sequence" branch: ego-pose decoder,
the two motion decoders, the TransDSSL depth decoder and the glue of `OneFormer.forward` around them.

TEST INFRASTRUCTURE: only `tests/` and `oracle/make_sequence_golden.py` import it.  Parity status: PINNED -- the generator runs the
reference's own modules (pose_decoder/resnet_like_pose_decoder.py, motion_decoder/dynamo_motion_decoder_mod.py,
pixel_decoder/transdssl.py, monodepth_loss.transformation_from_parameters; loaded by `oracle/ref_loader.load_sequence`) with
name-hashed weights and commits inputs + outputs as tests/golden/sequence_*.npz; tests/test_sequence_cpu.py checks this file
against them.  Citations are relative to /root/reference/model/.  State dict `sd`: name -> tensor, the reference's names.
"""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
SWIN_T_CH = (96, 192, 384, 768)


def _conv(x: Tensor, sd: SD, p: str, stride: int = 1, padding: int = 0) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _bn(x: Tensor, sd: SD, p: str, eps: float = 1e-5) -> Tensor:
    """eval-mode BatchNorm2d."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def residual_block(x: Tensor, sd: SD, p: str, stride: int, act) -> Tensor:
    """modeling/pose_decoder/resnet_like_pose_decoder.py:7-28 (act = relu) / motion_decoder/dynamo_motion_decoder_mod.py:5-27 (act = elu)."""
    out = F.relu(_bn(_conv(x, sd, p + ".left.0", stride, 1), sd, p + ".left.1"))
    out = _bn(_conv(out, sd, p + ".left.3", 1, 1), sd, p + ".left.4")
    sc = _bn(_conv(x, sd, p + ".shortcut.0", stride, 0), sd, p + ".shortcut.1") if (p + ".shortcut.0.weight") in sd else x
    return act(out + sc)


def _fusion_layer(x: Tensor, sd: SD, p: str, stride: int, act) -> Tensor:
    """make_layer / _make_fusion_layer: 1x1 conv, then two residual blocks (the first strided)."""
    x = _conv(x, sd, p + ".0")
    x = residual_block(x, sd, p + ".1", stride, act)
    return residual_block(x, sd, p + ".2", 1, act)


def resnet_like(features: Dict[str, Tensor], sd: SD, p: str = "pose_decoder", frames: int = 2) -> Tuple[Tensor, Tensor]:
    """pose_decoder/resnet_like_pose_decoder.py:52-72: features of the CONCATENATED (previous, current) backbone maps -> axis-angle
    and translation (B, frames, 1, 3) each, scaled by 0.01."""
    out = _fusion_layer(features["res2"], sd, p + ".layer1", 2, F.relu)
    out = _fusion_layer(torch.cat([out, features["res3"]], 1), sd, p + ".layer2", 2, F.relu)
    out = _fusion_layer(torch.cat([out, features["res4"]], 1), sd, p + ".layer3", 2, F.relu)
    out = _fusion_layer(torch.cat([out, features["res5"]], 1), sd, p + ".layer4", 2, F.relu)
    out = F.relu(_conv(out, sd, p + ".squeeze"))
    out = F.relu(_conv(out, sd, p + ".convs.pose_0", 1, 1))
    out = F.relu(_conv(out, sd, p + ".convs.pose_1", 1, 1))
    out = _conv(out, sd, p + ".convs.pose_2")
    out = 0.01 * out.mean(3).mean(2).view(-1, frames, 1, 6)
    return out[..., :3], out[..., 3:]


def motion_decoder_v2(motion_input: Dict[str, Tensor], ego_motion: Tensor, sd: SD, p: str, out_dim: int, scales=range(4)) -> Dict[tuple, Tensor]:
    """motion_decoder/dynamo_motion_decoder_mod.py:66-126.  motion_input: "full_res_input" (B, 6, H, W) + "res2".."res5"."""
    feat0 = motion_input["full_res_input"]
    feat1 = F.interpolate(motion_input["res2"], scale_factor=2, mode="bilinear", align_corners=False)
    feat1 = _fusion_layer(feat1, sd, p + ".layer0", 1, F.elu)
    field = _conv(100 * ego_motion, sd, p + ".res_trans_conv")
    outs = {}
    prev = field
    for s, feat in ((5, motion_input["res5"]), (4, motion_input["res4"]), (3, motion_input["res3"]), (2, motion_input["res2"]), (1, feat1), (0, feat0)):
        mf = F.interpolate(prev, size=feat.shape[-2:], mode="bilinear", align_corners=False)
        xa = _conv(torch.cat([mf, feat], 1), sd, f"{p}.conv{s}.0", 1, 1)
        xb = F.relu(_conv(xa, sd, f"{p}.conv{s}.1", 1, 1))
        prev = _conv(torch.cat([xa, xb], 1), sd, f"{p}.squeeze{s}") + mf
        outs[s] = prev
    res = {}
    for s in scales:
        if out_dim == 1:
            res[("motion_prob", s)] = 0.005 * outs[s]
            res[("motion_mask", s)] = torch.sigmoid(0.005 * outs[s])
        else:
            res[("complete_flow", s)] = 0.005 * outs[s]
    return res


def _rcu(x: Tensor, sd: SD, p: str) -> Tensor:
    """transdssl.py:114-184 ResidualConvUnit with use_norm False: relu -> conv3x3 -> relu -> conv3x3, + x."""
    out = _conv(F.relu(x), sd, p + ".conv1", 1, 1)
    out = _conv(F.relu(out), sd, p + ".conv2", 1, 1)
    return out + x


def _fusion(sd: SD, p: str, *xs: Tensor) -> Tensor:
    """transdssl.py:225-305 FeatureFusionBlock_custom (align_corners True)."""
    if len(xs) == 2:
        res = xs[0] + xs[1]
        att = F.softmax(_conv(_rcu(xs[1], sd, p + ".resConfUnit1"), sd, p + ".en_atten"), dim=1)
        out = _rcu(res * att, sd, p + ".resConfUnit2") + res
    else:
        out = _rcu(xs[0], sd, p + ".resConfUnit2")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(out, sd, p + ".out_conv")


def soft_att_depth(x: Tensor, alpha: float = 0.01, beta: float = 1.0) -> Tensor:
    """transdssl.py:187-222 ('UD'): expectation of linspace(alpha, beta, C) under the channel softmax."""
    grid = torch.linspace(alpha, beta, x.shape[1]).view(1, -1, 1, 1)
    return (F.softmax(x, dim=1) * grid).sum(1, keepdim=True)


def transdssl(features: Dict[str, Tensor], sd: SD, p: str = "sem_seg_head.depth_decoder") -> Dict[tuple, Tensor]:
    """transdssl.py:369-404 forward_features."""
    L = p + ".layers"
    l1, l2, l3, l4 = (_conv(features[f"res{i + 2}"], sd, f"{L}.layer{i + 1}_rn") for i in range(4))
    disp = lambda head, x: soft_att_depth(_conv(_conv(x, sd, f"{L}.{head}.0", 1, 1), sd, f"{L}.{head}.1", 1, 1))
    path_4 = _fusion(sd, L + ".refinenet4", l4)
    path_3 = _fusion(sd, L + ".refinenet3", path_4, l3)
    d3 = disp("output_conv4", path_3)
    path_2 = _fusion(sd, L + ".refinenet2", path_3, l2)
    d2 = disp("output_conv3", path_2)
    path_1 = _fusion(sd, L + ".refinenet1", path_2, l1)
    d1 = disp("output_conv2", path_1)
    l0 = F.interpolate(l1, scale_factor=2, mode="bilinear", align_corners=True)
    path_0 = _fusion(sd, L + ".refinenet0", path_1, l0)
    d0 = disp("output_conv", path_0)
    return {("disp", 3): d3, ("disp", 2): d2, ("disp", 1): d1, ("disp", 0): d0}


def rot_from_axisangle(vec: Tensor) -> Tensor:
    """monodepth_loss.py:187-224: (B, 1, 3) axis-angle -> (B, 4, 4)."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs, xC, yC, zC = x * sa, y * sa, z * sa, x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rot = torch.zeros((vec.shape[0], 4, 4), device=vec.device)
    rot[:, 0, 0] = torch.squeeze(x * xC + ca); rot[:, 0, 1] = torch.squeeze(xyC - zs); rot[:, 0, 2] = torch.squeeze(zxC + ys)
    rot[:, 1, 0] = torch.squeeze(xyC + zs); rot[:, 1, 1] = torch.squeeze(y * yC + ca); rot[:, 1, 2] = torch.squeeze(yzC - xs)
    rot[:, 2, 0] = torch.squeeze(zxC - ys); rot[:, 2, 1] = torch.squeeze(yzC + xs); rot[:, 2, 2] = torch.squeeze(z * zC + ca)
    rot[:, 3, 3] = 1
    return rot


def transformation_from_parameters(axisangle: Tensor, translation: Tensor, invert: bool = False) -> Tensor:
    """monodepth_loss.py:151-185."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = torch.zeros(t.shape[0], 4, 4, device=t.device)
    T[:, 0, 0] = T[:, 1, 1] = T[:, 2, 2] = T[:, 3, 3] = 1
    T[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def sequence_forward(cur: Tensor, prev: Tensor, feats_cur: Dict[str, Tensor], feats_prev: Dict[str, Tensor], sd: SD) -> dict:
    """oneformer_model.py:306-365 after the two backbone passes: cur / prev are the NORMALISED, padded image batches (B, 3, H, W)."""
    f_m = {k: torch.cat([feats_prev[k], feats_cur[k]], 1) for k in feats_cur}
    axis, trans = resnet_like(f_m, sd)
    axis, trans = axis[:, 0], trans[:, 0]
    cam = transformation_from_parameters(axis, trans, invert=True)
    motion_input = {"full_res_input": torch.cat([prev, cur], 1), **f_m}
    ego = torch.cat((trans.detach(), axis.detach()), -1).permute(0, 2, 1).unsqueeze(3)
    flow = motion_decoder_v2(motion_input, ego, sd, "motion_decoder", 3)
    mask = motion_decoder_v2(motion_input, ego, sd, "motion_mask", 1)
    disp = transdssl(feats_cur, sd)
    return {"disp_results": disp[("disp", 0)], "motion_mask": mask[("motion_mask", 0)], "complete_flow": flow[("complete_flow", 0)],
            "cam_T_cam": cam, "axisangle": axis, "translation": trans}


# ---------------------------------------------------------------------------------------------------------------------
# parameter / buffer shapes (state-dict names of the reference's modules)
# ---------------------------------------------------------------------------------------------------------------------
def _bn_shapes(p: str, c: int) -> Dict[str, tuple]:
    return {p + ".weight": (c,), p + ".bias": (c,), p + ".running_mean": (c,), p + ".running_var": (c,), p + ".num_batches_tracked": ()}


def _block_shapes(p: str, c: int, stride: int) -> Dict[str, tuple]:
    s = {p + ".left.0.weight": (c, c, 3, 3), **_bn_shapes(p + ".left.1", c), p + ".left.3.weight": (c, c, 3, 3), **_bn_shapes(p + ".left.4", c)}
    if stride != 1:
        s.update({p + ".shortcut.0.weight": (c, c, 1, 1), **_bn_shapes(p + ".shortcut.1", c)})
    return s


def _layer_shapes(p: str, cin: int, cout: int, stride: int) -> Dict[str, tuple]:
    return {p + ".0.weight": (cout, cin, 1, 1), p + ".0.bias": (cout,), **_block_shapes(p + ".1", cout, stride), **_block_shapes(p + ".2", cout, 1)}


def sequence_param_shapes() -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    p = "pose_decoder"
    for i, (ci, co) in enumerate(((192, 64), (384 + 64, 128), (768 + 128, 256), (1536 + 256, 512))):
        s.update(_layer_shapes(f"{p}.layer{i + 1}", ci, co, 2))
    s.update({p + ".squeeze.weight": (256, 512, 1, 1), p + ".squeeze.bias": (256,), p + ".convs.pose_0.weight": (256, 256, 3, 3), p + ".convs.pose_0.bias": (256,),
              p + ".convs.pose_1.weight": (256, 256, 3, 3), p + ".convs.pose_1.bias": (256,), p + ".convs.pose_2.weight": (12, 256, 1, 1), p + ".convs.pose_2.bias": (12,)})
    for p, od in (("motion_decoder", 3), ("motion_mask", 1)):
        for name, ci, co, st in (("layer0", 192, 64, 1), ("layer1", 64, 64, 2), ("layer2", 256, 64, 2), ("layer3", 448, 128, 2), ("layer4", 896, 256, 2)):
            s.update(_layer_shapes(f"{p}.{name}", ci, co, st))
        for st, n in enumerate((6, 64, 192, 384, 768, 1536)):
            s.update({f"{p}.conv{st}.0.weight": (n, n + od, 3, 3), f"{p}.conv{st}.0.bias": (n,), f"{p}.conv{st}.1.weight": (n, n, 3, 3), f"{p}.conv{st}.1.bias": (n,),
                      f"{p}.squeeze{st}.weight": (od, 2 * n, 1, 1), f"{p}.squeeze{st}.bias": (od,)})
        s.update({p + ".res_trans_conv.weight": (od, 6, 1, 1), p + ".res_trans_conv.bias": (od,)})
    L = "sem_seg_head.depth_decoder.layers"
    for i, c in enumerate(SWIN_T_CH):
        s[f"{L}.layer{i + 1}_rn.weight"] = (256, c, 1, 1)
    for i, n in ((0, 2), (1, 2), (2, 2), (3, 2), (4, 1)):
        r = f"{L}.refinenet{i}"
        s.update({r + ".out_conv.weight": (256, 256, 1, 1), r + ".out_conv.bias": (256,)})
        units = ["resConfUnit2"] + (["resConfUnit1"] if n == 2 else [])
        for u in units:
            for cv in ("conv1", "conv2"):
                s.update({f"{r}.{u}.{cv}.weight": (256, 256, 3, 3), f"{r}.{u}.{cv}.bias": (256,)})
        if n == 2:
            s.update({r + ".en_atten.weight": (256, 256, 1, 1), r + ".en_atten.bias": (256,)})
    for h in ("output_conv4", "output_conv3", "output_conv2", "output_conv"):
        s.update({f"{L}.{h}.0.weight": (128, 256, 3, 3), f"{L}.{h}.0.bias": (128,), f"{L}.{h}.1.weight": (32, 128, 3, 3), f"{L}.{h}.1.bias": (32,)})
    return s
