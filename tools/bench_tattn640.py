#!/usr/bin/env python3
"""Level-1 temporal attention (C = 640): insv2v_tattn_attn (LayerNorm + q/k/v + attention in one launch) vs row-linear q/k/v + insv2v_attention."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
from insv2v import ops
from insv2v.fused import pack_tattn_qkv_stream, pack_linear_stream
dev = torch.device("cuda:0")
C, H, F_, D, HW = 640, 8, 16, 80, 384
g = torch.Generator().manual_seed(0)
wqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).half()
table = torch.randn(F_, 3 * C, generator=g) * 0.4
st = pack_tattn_qkv_stream(wqkv.float(), table).to(dev)
sq = pack_linear_stream(wqkv.float(), None, table).to(dev)

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for samples in (3, 15, 30):
    M = samples * F_ * HW
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).half().to(dev)
    out, a2 = torch.empty_like(x), torch.empty_like(x)
    def fused(): ops.tattn_attn(x, st, samples, HW, H, F_, out=out)
    def split():
        qkv = ops.rowlin(x, sq, 3 * C, layernorm=True, frames=F_, rows_per_frame=HW)
        pq = qkv.data_ptr(); addr = (HW, F_ * HW * 3 * C, 3 * C)
        ops.attention(pq, pq + 2 * C, pq + 4 * C, a2, batch=samples * HW, heads=H, head_dim=D, seq_q=F_, seq_k=F_, scale=D ** -0.5,
                      q_rs=HW * 3 * C, k_rs=HW * 3 * C, v_rs=HW * 3 * C, o_rs=HW * C, q_addr=addr, kv_addr=addr, o_addr=(HW, F_ * HW * C, C))
    fused(); split(); torch.cuda.synchronize()
    print("max |fused - 2 launches| =", (out.float() - a2.float()).abs().max().item())
    flops = 2.0 * M * C * 3 * C + 4.0 * M * F_ * C
    for r in range(2):
        tf, ts = timeit(fused), timeit(split)
        print(f"samples={samples:2d} M={M:7d} round {r}: fused {tf:8.1f} us = {flops / tf * 1e-6:6.1f} TF/s | 2 launches {ts:8.1f} us = {flops / ts * 1e-6:6.1f} TF/s", flush=True)
