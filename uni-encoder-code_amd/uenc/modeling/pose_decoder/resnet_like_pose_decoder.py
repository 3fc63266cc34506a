"""Ego-pose decoder `ResNetLike` of the "sequence" branch -- counterpart of reference
model/modeling/pose_decoder/resnet_like_pose_decoder.py:7-72 (same module tree and state-dict names: `layer{1-4}.{0,1,2}`, `squeeze`,
`convs.pose_{0,1,2}`; channel counts hard-wired to the concatenated (previous, current) Swin-T features, :33-36).
Every convolution runs as a HIP GEMM with BatchNorm / bias / ReLU in its epilogue (uenc/convnet.py); eval mode only."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...convnet import conv, conv_bn_act


class ResidualBlock(nn.Module):
    def __init__(self, inchannel, outchannel, stride=1):
        super().__init__()
        self.left = nn.Sequential(nn.Conv2d(inchannel, outchannel, kernel_size=3, stride=stride, padding=1, bias=False),
                                  nn.BatchNorm2d(outchannel), nn.ReLU(inplace=True),
                                  nn.Conv2d(outchannel, outchannel, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(outchannel))
        self.shortcut = nn.Sequential()
        if stride != 1 or inchannel != outchannel:
            self.shortcut = nn.Sequential(nn.Conv2d(inchannel, outchannel, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(outchannel))
    act = staticmethod(F.relu)

    def forward(self, x):
        out = conv_bn_act(x, self.left[0], self.left[1], relu=True, out_dtype=torch.bfloat16)
        out = conv_bn_act(out, self.left[3], self.left[4])
        sc = conv_bn_act(x, self.shortcut[0], self.shortcut[1]) if len(self.shortcut) else x.float()
        return self.act(out + sc)


class ResNetLike(nn.Module):
    def __init__(self, ResidualBlock=ResidualBlock, num_input_features=1, num_frames_to_predict_for=2):
        super().__init__()
        self.layer1 = self.make_layer(ResidualBlock, 192, 64, 2, stride=2)
        self.layer2 = self.make_layer(ResidualBlock, 384 + 64, 128, 2, stride=2)
        self.layer3 = self.make_layer(ResidualBlock, 768 + 128, 256, 2, stride=2)
        self.layer4 = self.make_layer(ResidualBlock, 1536 + 256, 512, 2, stride=2)
        self.squeeze = nn.Conv2d(512, 256, 1)
        self.relu = nn.ReLU()
        self.convs = nn.ModuleDict({"pose_0": nn.Conv2d(num_input_features * 256, 256, 3, 1, 1), "pose_1": nn.Conv2d(256, 256, 3, 1, 1),
                                    "pose_2": nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)})
        self.num_frames_to_predict_for = num_frames_to_predict_for

    def make_layer(self, block, in_channels, out_channels, num_blocks, stride):
        layers = [nn.Conv2d(in_channels, out_channels, 1)]
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(out_channels, out_channels, s))
        return nn.Sequential(*layers)

    @staticmethod
    def _run(seq, x):
        x = conv(x, seq[0])
        for blk in list(seq)[1:]:
            x = blk(x)
        return x

    def forward(self, features):
        res2, res3, res4, res5 = features["res2"], features["res3"], features["res4"], features["res5"]
        out = self._run(self.layer1, res2)
        out = self._run(self.layer2, torch.cat([out, res3.float()], dim=1))
        out = self._run(self.layer3, torch.cat([out, res4.float()], dim=1))
        out = self._run(self.layer4, torch.cat([out, res5.float()], dim=1))
        out = conv(out, self.squeeze, relu=True)
        out = conv(out, self.convs["pose_0"], relu=True)
        out = conv(out, self.convs["pose_1"], relu=True)
        out = conv(out, self.convs["pose_2"])
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]
