"""warp_image / resize_flow: drop-in for misc_utils/flow_utils.py:25-57 and :59-86 on the HIP kernels.
RAFTFlow (flow_utils.py:134-189, torchvision pretrained) is out of scope: flows are supplied by the caller."""
import torch

from . import ops


def _dev32(t):
    if not t.is_cuda:
        t = t.cuda()
    return t.to(torch.float32)


def warp_image(image, flow, mode="bilinear"):
    """image (N,C,H,W) [or (C,H,W)], flow (N,2,H,W) -> bilinear sample of image at pixel + flow
    (grid_sample, align_corners=True, zero padding)."""
    if mode != "bilinear":
        raise NotImplementedError(mode)
    if image.dim() == 3:
        image = image.unsqueeze(0)
    if flow.dim() == 3:
        flow = flow.unsqueeze(0)
    assert image.shape[0] == flow.shape[0], \
        f"Batch size of image and flow must be the same. Got {image.shape[0]} and {flow.shape[0]}."
    assert image.shape[2:] == flow.shape[2:], \
        f"Height and width of image and flow must be the same. Got {image.shape[2:]} and {flow.shape[2:]}."
    return ops.warp_image(_dev32(image), _dev32(flow))


def resize_flow(flow, size):
    """Scale (u,v) by the resize factors, then bilinear resize (align_corners=False) to size=(H,W)."""
    return ops.resize_flow(_dev32(flow), tuple(int(s) for s in size))
