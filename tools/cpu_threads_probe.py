import sys, time, os, torch
sys.path[:0] = ['.', 'instruct-video-to-video_amd']
from insv2v import synth
import oracle.unet3d as ou
torch.set_grad_enabled(False)
m = ou.UNet3DConditionModel(**synth.UNET_FULL).eval()
x = torch.randn(1, 8, 16, 32, 48); ctx = torch.randn(1, 77, 768)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    t0 = time.perf_counter(); m(x, torch.tensor([981]), ctx); print(th, 'threads:', round(time.perf_counter() - t0, 1), 's', flush=True)
