"""Pose normalisation constants and the 6D rotation helpers of the regression head.

Mirrors mp3d_loftr/src/losses/loftr_loss.py:7-8 (pose_mean_6d / pose_std_6d: dataset statistics, data),
:10-29 (rotation_6d_to_matrix), :31-39 (matrix_to_rotation_6d, compute_normalized_6d)."""
import torch
import torch.nn.functional as F

pose_mean_6d = torch.tensor([-0.34898765, 0.17085525, -0.87944315, 0.50275223, 0.03533648, -0.18179045,
                             -0.03533648, 0.98189617, 0.09313615])
pose_std_6d = torch.tensor([1.94014405, 0.36770130, 1.88317520, 0.51837117, 0.12717603, 0.65426397,
                            0.12717603, 0.0188729, 0.09709263])


def rotation_6d_to_matrix(d6):
    """Gram-Schmidt of the two 3-vectors in d6 (..., 6) -> rotation matrices (..., 3, 3), rows b1, b2, b3."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def matrix_to_rotation_6d(R):
    return R[..., :2, :].clone().reshape(*R.size()[:-2], 6)


def compute_normalized_6d(pose):
    """pose (..., 3, 4) -> (..., 9): [t, first two rows of R] standardised with the dataset statistics."""
    v = torch.cat([pose[..., :3, 3], matrix_to_rotation_6d(pose[..., :3, :3])], dim=-1)
    return (v - pose_mean_6d.to(v.device)) / pose_std_6d.to(v.device)
