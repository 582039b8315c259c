"""CPU restatement of misc_utils/flow_utils.py:25-57 (warp_image) and :59-86
(resize_flow) (test oracle)."""
import torch
import torch.nn.functional as F


def warp_image(image, flow, mode="bilinear"):
    if image.dim() == 3:
        image = image.unsqueeze(0)
    if flow.dim() == 3:
        flow = flow.unsqueeze(0)
    assert image.shape[0] == flow.shape[0] and image.shape[2:] == flow.shape[2:]
    n, _, h, w = image.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    gx = xs[None] + flow[:, 0]
    gy = ys[None] + flow[:, 1]
    grid = torch.stack([2 * (gx / (w - 1) - 0.5), 2 * (gy / (h - 1) - 0.5)], dim=-1)
    return F.grid_sample(image, grid, mode=mode, align_corners=True)


def resize_flow(flow, size):
    H, W = size
    h, w = flow.shape[2:]
    scaled = flow.clone()
    scaled[:, 0] *= W / w
    scaled[:, 1] *= H / h
    return F.interpolate(scaled, size=(H, W), mode="bilinear", align_corners=False)
