"""RAFTFlow on the HIP kernels: drop-in for misc_utils/flow_utils.py:134-189 (the estimator the reference builds at
pl_trainer/inference/inference.py:294 and calls once per query frame at :303-311).

The network is torchvision's ``raft_large`` (third party, not under the reference tree and not installed here: the architecture is
restated from the published model, oracle/raft.py states what that pins and what it cannot).  ``load_state_dict`` takes torchvision's
own key names (``feature_encoder.*``, ``context_encoder.*``, ``update_block.*``, ``mask_predictor.*``), so the real
``raft_large_C_T_SKHT_V2`` checkpoint is a drop-in; offline the weights are key-hashed (insv2v/synth.py).

Layout: activations are channels-last fp16 token matrices [n*h*w, C]; every convolution is ``ops.im2col`` + ``ops.gemm`` (bias and
ReLU / sigmoid / tanh in the GEMM epilogue, eval-mode BatchNorm folded into the weights); InstanceNorm, the residual adds, the ConvGRU
blends, the correlation pyramid, its 4 x 81 bilinear look-ups per pixel and the convex upsampling are kernels of csrc/raft.hip.
Correspondences and flows stay fp32 [n, 2, h, w].  Runs once per window - HBM-bound gathers and small GEMMs, not the hot loop.
"""
import torch

from . import ops

CPAD = 8   # image / flow channels are zero-padded to one 16-byte chunk
import os
IMPLICIT_CONV3X3 = os.environ.get("INSV2V_RAFT_IMPLICIT_CONV", "1") != "0"   # A/B switch: 3x3 layers without im2col


def _dev(t, dtype, device):
    return t.detach().to(device=device, dtype=dtype).contiguous()


class _Conv:
    """One Conv2d as a GEMM weight [Cout_pad, kh*kw*Cin_pad] (K order = tap-major, channel-minor: ops.im2col's) + fp32 bias."""

    def __init__(self, sd, key, device, bn_key=None, cin_pad=None, cout_pad=None, scale=1.0, bn_eps=1e-5):
        w = sd[key + ".weight"].detach().float()
        b = sd[key + ".bias"].detach().float()
        if bn_key is not None:   # eval-mode BatchNorm2d folded in: y = (conv(x) - mean) * gamma / sqrt(var + eps) + beta
            g = sd[bn_key + ".weight"].float() / torch.sqrt(sd[bn_key + ".running_var"].float() + bn_eps)
            w, b = w * g[:, None, None, None], (b - sd[bn_key + ".running_mean"].float()) * g + sd[bn_key + ".bias"].float()
        w, b = w * scale, b * scale
        cout, cin, self.kh, self.kw = w.shape
        self.cin = cin_pad or (cin + 7) // 8 * 8
        self.cout = cout
        n = cout_pad or (cout + 7) // 8 * 8
        wk = torch.zeros(n, self.kh, self.kw, self.cin)
        wk[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        bk = torch.zeros(n)
        bk[:cout] = b
        self.w, self.b = _dev(wk.reshape(n, -1), torch.float16, device), _dev(bk, torch.float32, device)

    def __call__(self, x, geom, stride=1, act=ops.ACT_NONE, out=None, x2=None, out_fp32=False):
        pad = ((self.kh - 1) // 2, (self.kw - 1) // 2)
        if self.kh == 1 and self.kw == 1 and stride == 1 and x2 is None:
            cols, g = x, geom
        elif (IMPLICIT_CONV3X3 and self.kh == 3 and self.kw == 3 and self.cin % 64 == 0 and (x2 is None or x.shape[1] % 64 == 0)
              and x.shape[1] + (x2.shape[1] if x2 is not None else 0) == self.cin):
            # round 6: the 3x3 layers with whole 64-channel K slices run as the implicit-GEMM convolution of insv2v_gemm
            # (bias + ReLU in its epilogue) instead of materialising [rows, 9 * Cin] with im2col - 41 % of the estimator's time went there
            return ops.conv3x3(x, geom, self.w, self.b, x2=x2, stride=stride, pad=pad, act=act, out_fp32=out_fp32, out=out)
        else:
            cols, g = ops.im2col(x, geom, self.cin, self.kh, self.kw, stride, pad, x2=x2)
        return ops.gemm(cols, self.w, self.b, act=act, out=out, out_fp32=out_fp32), g


class _ResBlock:
    def __init__(self, sd, key, device, norm, stride):
        bn = (lambda k: k + ".1") if norm == "batch" else (lambda k: None)
        self.norm, self.stride = norm, stride
        self.c1 = _Conv(sd, key + ".convnormrelu1.0", device, bn(key + ".convnormrelu1"))
        self.c2 = _Conv(sd, key + ".convnormrelu2.0", device, bn(key + ".convnormrelu2"))
        self.down = _Conv(sd, key + ".downsample.0", device, bn(key + ".downsample")) if stride != 1 else None

    def _cnr(self, conv, x, geom, stride, relu=True):
        if self.norm == "batch":
            return conv(x, geom, stride, act=ops.ACT_RELU if relu else ops.ACT_NONE)
        y, g = conv(x, geom, stride)
        return ops.instance_norm(y, g[0], g[1] * g[2], relu=relu), g

    def __call__(self, x, geom):
        y, g = self._cnr(self.c1, x, geom, self.stride)
        y, g = self._cnr(self.c2, y, g, 1)
        if self.down is not None:
            x, _ = self._cnr(self.down, x, geom, self.stride, relu=False)
        return ops.ew(ops.EW_ADD_RELU, x, y), g


class _Encoder:
    """torchvision FeatureEncoder: conv7x7 s2 -> norm -> ReLU -> 3 x 2 ResidualBlocks (strides 1, 2, 2) -> conv1x1."""

    def __init__(self, sd, key, device, norm):
        self.norm = norm
        self.stem = _Conv(sd, key + ".convnormrelu.0", device, (key + ".convnormrelu.1") if norm == "batch" else None, cin_pad=CPAD)
        self.blocks = []
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            self.blocks += [_ResBlock(sd, f"{key}.layer{li}.0", device, norm, stride), _ResBlock(sd, f"{key}.layer{li}.1", device, norm, 1)]
        self.conv = _Conv(sd, key + ".conv", device)

    def __call__(self, x, geom):
        if self.norm == "batch":
            y, g = self.stem(x, geom, 2, act=ops.ACT_RELU)
        else:
            y, g = self.stem(x, geom, 2)
            y = ops.instance_norm(y, g[0], g[1] * g[2], relu=True)
        for blk in self.blocks:
            y, g = blk(y, g)
        return self.conv(y, g)


class RAFT:
    """torchvision ``raft_large``: feature / context encoders, 4-level correlation pyramid (radius 4), ConvGRU update block,
    convex 8x upsampling.  ``__call__(image1, image2)`` -> list with the LAST flow prediction [B, 2, H, W] (all RAFTFlow consumes)."""
    LEVELS, RADIUS, HIDDEN, CONTEXT = 4, 4, 128, 128

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.loaded = False

    def load_state_dict(self, sd, strict=True):
        dev = self.device
        self.fnet = _Encoder(sd, "feature_encoder", dev, "instance")
        self.cnet = _Encoder(sd, "context_encoder", dev, "batch")
        m = "update_block.motion_encoder"
        ncorr = self.LEVELS * (2 * self.RADIUS + 1) ** 2                    # 324 look-ups per pixel
        self.corr_ld = (ncorr + 7) // 8 * 8
        self.convcorr1 = _Conv(sd, m + ".convcorr1.0", dev, cin_pad=self.corr_ld)
        self.convcorr2 = _Conv(sd, m + ".convcorr2.0", dev)
        self.convflow1 = _Conv(sd, m + ".convflow1.0", dev, cin_pad=CPAD)
        self.convflow2 = _Conv(sd, m + ".convflow2.0", dev)
        self.convmotion = _Conv(sd, m + ".conv.0", dev)                    # 126 outputs, padded to 128: the flow fills the last two
        g = "update_block.recurrent_block"
        self.gru = []
        for name in ("convgru1", "convgru2"):
            zr = {"weight": torch.cat([sd[f"{g}.{name}.convz.weight"], sd[f"{g}.{name}.convr.weight"]], 0),
                  "bias": torch.cat([sd[f"{g}.{name}.convz.bias"], sd[f"{g}.{name}.convr.bias"]], 0)}
            self.gru.append((_Conv({"zr.weight": zr["weight"], "zr.bias": zr["bias"]}, "zr", dev), _Conv(sd, f"{g}.{name}.convq", dev)))
        self.flow1 = _Conv(sd, "update_block.flow_head.conv1", dev)
        self.flow2 = _Conv(sd, "update_block.flow_head.conv2", dev)
        self.mask1 = _Conv(sd, "mask_predictor.convrelu.0", dev)
        self.mask2 = _Conv(sd, "mask_predictor.conv", dev, scale=0.25)     # MaskPredictor.multiplier
        self.loaded = True
        return self

    @torch.no_grad()
    def __call__(self, image1, image2, num_flow_updates=12):
        if not self.loaded:
            raise RuntimeError("RAFT: load_state_dict() has not been called")
        B, _, H, W = image1.shape
        if image2.shape[-2:] != (H, W) or H % 8 or W % 8:
            raise ValueError("input images must share a shape divisible by 8")
        h, w = H // 8, W // 8
        if (h >> (self.LEVELS - 1)) < 2 or (w >> (self.LEVELS - 1)) < 2:
            raise ValueError("images too small for the 4-level correlation pyramid (the reference divides by zero there)")
        dev = self.device
        imgs = torch.cat([image1, image2], 0).to(device=dev, dtype=torch.float32).contiguous()
        x = ops.nchw_to_nhwc_f16(imgs, CPAD)
        fmaps, _ = self.fnet(x, (2 * B, H, W))                              # [2B*h*w, 256]
        rows = h * w
        C = fmaps.shape[1]
        # all-pairs correlation: fp32 [B, hw, hw] = fmap1 . fmap2^T / sqrt(C), then the pooled pyramid over fmap2's (h, w)
        corr = torch.empty((B, rows, rows), device=dev, dtype=torch.float32)
        ops.gemm(fmaps[:B * rows], fmaps[B * rows:], out=corr, out_fp32=True, alpha=C ** -0.5, batch=B, M=rows, N=rows, K=C,
                 a_bs=rows * C, w_bs=rows * C, c_bs=rows * rows, lda=C, ldw=C, ldc=rows)
        pyramid = [corr.reshape(B * rows, h, w)]
        for lv in range(1, self.LEVELS):
            pyramid.append(ops.avgpool2x2(pyramid[-1], B * rows, h >> (lv - 1), w >> (lv - 1)))
        ctx_out, _ = self.cnet(x[:B * H * W], (B, H, W))                     # [B*h*w, 256]: hidden | context
        n = B * rows
        hx = torch.empty((n, self.HIDDEN + self.CONTEXT + 128), device=dev, dtype=torch.float16)   # [h | context | motion features]
        ops.ew(ops.EW_TANH, ctx_out[:, :self.HIDDEN], out=hx[:, :self.HIDDEN])
        ops.ew(ops.EW_RELU, ctx_out[:, self.HIDDEN:], out=hx[:, self.HIDDEN:self.HIDDEN + self.CONTEXT])
        hid, xin, motion = hx[:, :self.HIDDEN], hx[:, self.HIDDEN:], hx[:, self.HIDDEN + self.CONTEXT:]
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        coords1 = torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1).contiguous()
        flow_rows = torch.zeros((n, CPAD), device=dev, dtype=torch.float16)
        corrflow = torch.empty((n, 192 + 64), device=dev, dtype=torch.float16)
        rh = torch.empty((n, self.HIDDEN), device=dev, dtype=torch.float16)
        geom = (B, h, w)
        for _ in range(num_flow_updates):
            feats = ops.corr_lookup(pyramid, coords1, B, h, w, self.RADIUS, self.corr_ld)
            # MotionEncoder: corr 1x1 -> 3x3 | flow 7x7 -> 3x3 | cat -> 3x3 (126) | cat flow (2)
            c1, _ = self.convcorr1(feats, geom, act=ops.ACT_RELU)
            self.convcorr2(c1, geom, act=ops.ACT_RELU, out=corrflow[:, :192])
            f1, _ = self.convflow1(flow_rows, geom, act=ops.ACT_RELU)
            self.convflow2(f1, geom, act=ops.ACT_RELU, out=corrflow[:, 192:])
            self.convmotion(corrflow, geom, act=ops.ACT_RELU, out=motion)
            ops.raft_flow_rows(coords1, None, motion[:, 126:128], B, h, w)
            # RecurrentBlock: ConvGRU (1x5) then ConvGRU (5x1) on [h | x]
            for zr_conv, q_conv in self.gru:
                zr, _ = zr_conv(hx, geom, act=ops.ACT_SIGMOID)
                ops.ew(ops.EW_GRU_RH, zr[:, self.HIDDEN:], hid, out=rh)
                q, _ = q_conv(rh, geom, act=ops.ACT_TANH, x2=xin)
                ops.ew(ops.EW_GRU_OUT, q, hid, zr[:, :self.HIDDEN], out=hid)
            # FlowHead -> delta; coords1 += delta; flow rows for the next iteration
            d1, _ = self.flow1(hid, geom, act=ops.ACT_RELU)
            delta, _ = self.flow2(d1, geom, out_fp32=True)
            ops.raft_flow_rows(coords1, delta, flow_rows, B, h, w)
        m1, _ = self.mask1(hid, geom, act=ops.ACT_RELU)
        mask, _ = self.mask2(m1, geom)
        return [ops.convex_upsample(coords1, mask, B, h, w)]


IM2COL_BUDGET_BYTES = 1 << 30


class RAFTFlow:
    """flow_utils.py:134-189: ``flow = RAFTFlow()(img1, img2[, img_size])`` -> [B, 2, H, W], the LAST of the model's 12 predictions.
    The preset transform maps [0, 1] -> [-1, 1] (x -> 2 x - 1) whatever it is handed (:176), as in the reference."""

    # (query, reference) pairs the optical-flow pipe may hand over in one call (inference.obtain_flow_batched); __call__ runs them in
    # equal chunks whose widest im2col buffer (the 7x7 stem's, 38.5 MB per 256x384 pair) stays inside IM2COL_BUDGET_BYTES (two calls of 24)
    max_pairs = 48

    def __init__(self, device="cuda", state_dict=None):
        self.model = RAFT(device)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)

    def load_state_dict(self, sd, strict=True):
        self.model.load_state_dict({k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()})
        return self

    def cuda(self):
        return self

    @torch.no_grad()
    def __call__(self, img1, img2, img_size=None, num_flow_updates=12):
        original = tuple(img1.shape[2:])
        dev = self.model.device
        img1, img2 = img1.to(device=dev, dtype=torch.float32), img2.to(device=dev, dtype=torch.float32)
        if img_size is not None:
            raise NotImplementedError("RAFTFlow(img_size=...): the resize branch (flow_utils.py:171-174) is not used by the sampling path")
        img1, img2 = (img1 - 0.5) / 0.5, (img2 - 0.5) / 0.5
        # pairs per estimator call from a MEMORY budget, not from the 2 GiB operand window insv2v_gemm no longer needs respected.
        # The widest im2col buffer left (round 6: the 3x3 layers run as implicit-GEMM convolutions) is the 7x7 stem's: 2 images x
        # (H/2 * W/2) rows x 49 taps x 8 channels x 2 B = 38.5 MB per 256x384 pair -> 26 pairs per GiB: a window's 48 pairs run as two calls of 24
        cap = max(1, IM2COL_BUDGET_BYTES // (2 * (original[0] // 2) * (original[1] // 2) * 784))
        cap = -(-img1.shape[0] // -(-img1.shape[0] // cap))   # equal chunks
        flow = torch.cat([self.model(img1[i:i + cap], img2[i:i + cap], num_flow_updates)[-1] for i in range(0, img1.shape[0], cap)], 0)
        assert tuple(flow.shape[2:]) == original
        return flow

    forward = __call__
