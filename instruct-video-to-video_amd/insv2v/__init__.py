"""insv2v: MI355X-native InsV2V denoising hot path.

Importing the package never touches the GPU; kernels are loaded on first use and there is no CPU
fallback (instruct-video-to-video_amd/insv2v/_lib.py)."""
from . import synth  # noqa: F401

__all__ = ["UNet3DConditionModel", "AutoencoderKL", "InferenceIP2PVideo", "InferenceIP2PVideoOpticalFlow",
           "InstructP2PVideoModel", "create_model", "warp_image", "resize_flow", "split_batch"]


def __getattr__(name):
    import importlib
    table = {
        "UNet3DConditionModel": ".unet", "AutoencoderKL": ".vae", "InferenceIP2PVideo": ".inference",
        "InferenceIP2PVideoOpticalFlow": ".inference", "Inference": ".inference",
        "InstructP2PVideoModel": ".model", "create_model": ".model", "unit_test_create_model": ".model",
        "warp_image": ".flow_utils", "resize_flow": ".flow_utils", "split_batch": ".run_loveu_tgve",
        "edit_video": ".run_loveu_tgve",
    }
    if name in table:
        return getattr(importlib.import_module(table[name], __name__), name)
    raise AttributeError(name)
