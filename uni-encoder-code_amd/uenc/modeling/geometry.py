"""Pose algebra of the "sequence" branch: axis-angle + translation -> 4x4 camera transform (reference model/modeling/monodepth_loss.py
:151-224, used at oneformer_model.py:331-336).  Six numbers per image: plain tensor arithmetic, no kernel."""
import torch


def rot_from_axisangle(vec: torch.Tensor) -> torch.Tensor:
    """(B, 1, 3) -> (B, 4, 4) rotation (Rodrigues), monodepth_loss.py:187-224."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = axis[..., 0].unsqueeze(1), axis[..., 1].unsqueeze(1), axis[..., 2].unsqueeze(1)
    rows = [[x * x * C + ca, x * y * C - z * sa, z * x * C + y * sa],
            [x * y * C + z * sa, y * y * C + ca, y * z * C - x * sa],
            [z * x * C - y * sa, y * z * C + x * sa, z * z * C + ca]]
    rot = torch.zeros((vec.shape[0], 4, 4), device=vec.device, dtype=vec.dtype)
    for i in range(3):
        for j in range(3):
            rot[:, i, j] = rows[i][j].reshape(-1)
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(t: torch.Tensor) -> torch.Tensor:
    """(B, 1, 3) -> (B, 4, 4), monodepth_loss.py:171-185."""
    T = torch.eye(4, device=t.device, dtype=t.dtype).repeat(t.shape[0], 1, 1)
    T[:, :3, 3] = t.reshape(-1, 3)
    return T


def transformation_from_parameters(axisangle: torch.Tensor, translation: torch.Tensor, invert: bool = False) -> torch.Tensor:
    """monodepth_loss.py:151-168: T R, or (with invert) R^T T(-t) -- the inverse transform."""
    R = rot_from_axisangle(axisangle)
    t = translation
    if invert:
        R = R.transpose(1, 2)
        t = -t
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)
