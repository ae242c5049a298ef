"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's image losses (utils/loss_utils.py).

Pinned: tests/golden/loss_cases.npz holds values and autograd gradients produced by the reference's own `l1_loss` and
`ssim` (tools/gen_golden.py imports /root/reference/utils/loss_utils.py in the build container).
"""
import math

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    """utils/loss_utils.py:29"""
    return torch.abs(network_output - gt).mean()


def gaussian_window(window_size=11, sigma=1.5):
    """utils/loss_utils.py:46-48: exp(-(x - ws//2)^2 / (2 sigma^2)) in python floats -> float32, normalised in float32."""
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:59-86: depthwise conv2d with the 2D window w1d w1d^T, zero padding window_size // 2,
    C1 = 0.01^2, C2 = 0.03^2; (C,H,W) or (N,C,H,W) inputs."""
    channel = img1.size(-3)
    w1 = gaussian_window(window_size).unsqueeze(1)
    window = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()
    window = window.type_as(img1)
    pad = window_size // 2
    conv = lambda t: F.conv2d(t, window, padding=pad, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    if size_average:
        return ssim_map.mean()
    return ssim_map.mean(1).mean(1).mean(1)
