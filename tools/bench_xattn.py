#!/usr/bin/env python3
"""Level-0 text cross-attention sub-block: insv2v_xattn_fused vs row-linear q + insv2v_attention + row-linear out-projection."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]
import torch
from insv2v import ops
from insv2v.fused import pack_xattn_stream, pack_xattn_kv, pack_linear_stream
dev = torch.device("cuda:0")
C, H, D, L, rows = 320, 8, 40, 77, 16 * 1536
g = torch.Generator().manual_seed(0)
wq, bq = (torch.randn(C, C, generator=g) * C ** -0.5).half(), torch.randn(C, generator=g) * 0.3
wo, bo = (torch.randn(C, C, generator=g) * C ** -0.5).half(), torch.randn(C, generator=g) * 0.3
st = pack_xattn_stream(wq.float(), bq, wo.float(), bo).to(dev)
wo1, bo1 = (torch.randn(C, C, generator=g) * C ** -0.5).half(), torch.randn(C, generator=g) * 0.3
stp = pack_xattn_stream(wq.float(), bq, wo.float(), bo, pre=(wo1.float(), bo1)).to(dev)
s1 = pack_linear_stream(wo1.float(), bo1).to(dev)
sq, so = pack_linear_stream(wq.float(), bq).to(dev), pack_linear_stream(wo.float(), bo).to(dev)

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for samples in (3, 15, 30):
    M = samples * rows
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).half().to(dev)
    kv = (torch.randn(samples * L, 2 * C, generator=g) * 1.5).half().to(dev)
    kvs = pack_xattn_kv(kv, samples, L, C, H)
    out, a2 = torch.empty_like(x), torch.empty_like(x)
    def fused(): ops.xattn_fused(x, st, kvs, rows, H, L, out=out)
    def split():
        q = ops.rowlin(x, sq, C, layernorm=True)
        kp = kv.data_ptr()
        ops.attention(q.data_ptr(), kp, kp + 2 * C, a2, batch=samples * 16, heads=H, head_dim=D, seq_q=rows // 16, seq_k=L, scale=D ** -0.5,
                      q_rs=C, k_rs=2 * C, v_rs=2 * C, o_rs=C, q_addr=(1, rows // 16 * C, 0), kv_addr=(16, L * 2 * C, 0), o_addr=(1, rows // 16 * C, 0))
        ops.rowlin(a2, so, C, residual=x, out=out)
    fused(); r1 = out.clone(); split(); torch.cuda.synchronize()
    print("max |fused - 3 launches| =", (r1.float() - out.float()).abs().max().item())
    flops = 4.0 * M * C * C + 4.0 * M * L * C
    for r in range(2):
        tf, ts = timeit(fused), timeit(split)
        print(f"samples={samples:2d} M={M:7d} round {r}: fused {tf:8.1f} us = {flops / tf * 1e-6:6.1f} TF/s | 3 launches {ts:8.1f} us = {flops / ts * 1e-6:6.1f} TF/s", flush=True)
    a1 = (torch.randn(M, C, generator=g) * 0.8).half().to(dev)
    x1 = torch.empty_like(x)
    def pre(): ops.xattn_fused(a1, stp, kvs, rows, H, L, out=out, pre_residual=x)
    def two():
        ops.rowlin(a1, s1, C, residual=x, out=x1)
        ops.xattn_fused(x1, st, kvs, rows, H, L, out=out)
    for r in range(2):
        tp, t2 = timeit(pre), timeit(two)
        print(f"samples={samples:2d} M={M:7d} round {r}: out-proj + block in one launch {tp:8.1f} us | row-Linear + block {t2:8.1f} us", flush=True)
