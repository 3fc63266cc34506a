"""`MotionDecoderV2` of the "sequence" branch (complete 3-D flow with out_dim 3, motion mask with out_dim 1) -- counterpart of reference
model/modeling/motion_decoder/dynamo_motion_decoder_mod.py:30-126: same module tree / state-dict names (`layer{0-4}`, `conv{0-5}`,
`squeeze{0-5}`, `res_trans_conv`; `layer1..4` are constructed but never used by the reference's forward either, :40-43), same
coarse-to-fine refinement of the ego-motion field.  Convolutions on the HIP GEMMs (uenc/convnet.py); eval mode only."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...convnet import conv
from ..pose_decoder.resnet_like_pose_decoder import ResidualBlock as _ReluBlock


class ResidualBlock(_ReluBlock):
    """Same block with ELU after the residual sum (dynamo_motion_decoder_mod.py:22-26)."""
    act = staticmethod(F.elu)


class MotionDecoderV2(nn.Module):
    def __init__(self, scales=range(4), num_input_images=2, out_dim=4):
        super().__init__()
        self.num_inp_feat = [6, 64, 192, 384, 768, 1536]
        self.out_dim, self.scales = out_dim, scales
        self.layer0 = self._make_fusion_layer(ResidualBlock, 192, 64, 2, stride=1)
        self.layer1 = self._make_fusion_layer(ResidualBlock, 64, 64, 2, stride=2)
        self.layer2 = self._make_fusion_layer(ResidualBlock, 192 + 64, 64, 2, stride=2)
        self.layer3 = self._make_fusion_layer(ResidualBlock, 384 + 64, 128, 2, stride=2)
        self.layer4 = self._make_fusion_layer(ResidualBlock, 768 + 128, 256, 2, stride=2)
        for s in range(6):
            c, q = self._make_layer(s)
            setattr(self, f"conv{s}", c); setattr(self, f"squeeze{s}", q)
        self.res_trans_conv = nn.Conv2d(6, out_dim, kernel_size=1, stride=1, padding=0)
        self.softmax = nn.Softmax(dim=1)

    def _make_fusion_layer(self, block, in_channels, out_channels, num_blocks, stride):
        layers = [nn.Conv2d(in_channels, out_channels, 1)]
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(out_channels, out_channels, s))
        return nn.Sequential(*layers)

    def _make_layer(self, stage):
        n = self.num_inp_feat[stage]
        return (nn.Sequential(nn.Conv2d(n + self.out_dim, n, kernel_size=3, stride=1, padding=1), nn.Conv2d(n, n, kernel_size=3, stride=1, padding=1), nn.ReLU()),
                nn.Conv2d(n * 2, self.out_dim, kernel_size=1, stride=1))

    def _stage(self, s, motion_prev, feat):
        """One refinement step (:77-80 and its five repetitions): upsample the coarser field, two 3x3 convs on [field, features],
        1x1 squeeze of both conv outputs, residual on the field."""
        cv, sq = getattr(self, f"conv{s}"), getattr(self, f"squeeze{s}")
        field = F.interpolate(motion_prev, size=feat.shape[-2:], mode="bilinear", align_corners=False)
        xa = conv(torch.cat([field, feat.float()], dim=1), cv[0], out_dtype=torch.bfloat16)
        xb = conv(xa, cv[1], relu=True, out_dtype=torch.bfloat16)
        return conv(torch.cat([xa, xb], dim=1), sq) + field

    def forward(self, pose_feat, ego_motion):
        mi = pose_feat["motion_input"]
        feat0 = mi["full_res_input"]
        feat1 = F.interpolate(mi["res2"].detach().float(), scale_factor=2, mode="bilinear", align_corners=False)
        x = conv(feat1, self.layer0[0])
        for blk in list(self.layer0)[1:]:
            x = blk(x)
        feat1 = x
        res_trans = F.conv2d(100 * ego_motion, self.res_trans_conv.weight, self.res_trans_conv.bias)      # (B, out_dim, 1, 1): six inputs per image
        out5 = self._stage(5, res_trans, mi["res5"])
        out4 = self._stage(4, out5, mi["res4"])
        out3 = self._stage(3, out4, mi["res3"])
        out2 = self._stage(2, out3, mi["res2"])
        out1 = self._stage(1, out2, feat1)
        out0 = self._stage(0, out1, feat0)
        outs = {0: out0, 1: out1, 2: out2, 3: out3, 4: out4, 5: out5}
        outputs = {}
        for scale in self.scales:
            if self.out_dim == 1:
                outputs[("motion_prob", scale)] = 0.005 * outs[scale]
                outputs[("motion_mask", scale)] = torch.sigmoid(0.005 * outs[scale])
            elif self.out_dim == 3:
                outputs[("complete_flow", scale)] = 0.005 * outs[scale]
            else:
                raise Exception(f"out_dim={self.out_dim} not excepted.")
        return outputs
