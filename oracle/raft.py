"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU fp32 restatement of the optical-flow estimator behind
``RAFTFlow`` (reference: misc_utils/flow_utils.py:134-189, called at pl_trainer/inference/inference.py:294,303-311).

The arithmetic lives in a THIRD-PARTY dependency that is absent from /root/reference and from this image: torchvision
(``torchvision.models.optical_flow.raft_large`` + ``Raft_Large_Weights.DEFAULT.transforms()``; requirements.txt pins no
version).  PARITY UNPINNED: this file restates the published torchvision architecture (RAFT, Teed & Deng 2020, as packaged in
torchvision >= 0.12: models/optical_flow/raft.py, _utils.py, transforms/_presets.py ``OpticalFlow``) module by module with the
SAME state-dict key names, so the real ``raft_large_C_T_SKHT_V2`` checkpoint loads unchanged - but nothing in the container can
confirm the reading (no torchvision wheel, no golden vectors in the reference, no network for the checkpoint).

  FeatureEncoder / ResidualBlock      conv7x7 s2 -> 3 x 2 residual blocks (64, 96 s2, 128 s2) -> conv1x1 -> 256 ch at 1/8 resolution
                                      (InstanceNorm2d for the feature encoder, eval-mode BatchNorm2d for the context encoder)
  CorrBlock                           all-pairs correlation fmap1 . fmap2 / sqrt(256), 4-level average-pooled pyramid,
                                      (2 * 4 + 1)^2 bilinear look-ups per level around the current correspondence
  MotionEncoder / ConvGRU x 2 / FlowHead   the update block, 12 iterations (torchvision's default num_flow_updates)
  MaskPredictor + upsample_flow       convex 8x upsampling with a softmax over the 9 neighbours
  RAFTFlow.forward                    flow_utils.py:160-189: the preset maps [0, 1] -> [-1, 1] (x -> 2 x - 1) whatever range it is
                                      handed (the reference hands it frames that are ALREADY in [-1, 1]; restated as is), the LAST
                                      flow prediction is returned
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def conv_norm_act(cin, cout, norm, k, stride=1, act=True):
    """torchvision.ops.misc.Conv2dNormActivation: Sequential(conv[, norm][, ReLU]); padding (k - 1) // 2; the conv has a bias exactly when
    torchvision gives it one (bias=True is passed wherever a norm follows in raft.py; without a norm the default is bias=True too)."""
    pad = tuple((kk - 1) // 2 for kk in k) if isinstance(k, tuple) else (k - 1) // 2
    layers = [nn.Conv2d(cin, cout, k, stride, pad, bias=True)]
    if norm is not None:
        layers.append(norm(cout))
    if act:
        layers.append(nn.ReLU(inplace=False))
    return nn.Sequential(*layers)


class ResidualBlock(nn.Module):
    def __init__(self, cin, cout, norm, stride=1):
        super().__init__()
        self.convnormrelu1 = conv_norm_act(cin, cout, norm, 3, stride)
        self.convnormrelu2 = conv_norm_act(cout, cout, norm, 3)
        self.downsample = nn.Identity() if stride == 1 else conv_norm_act(cin, cout, norm, 1, stride, act=False)

    def forward(self, x):
        y = self.convnormrelu2(self.convnormrelu1(x))
        return F.relu(self.downsample(x) + y)


class FeatureEncoder(nn.Module):
    def __init__(self, norm, layers=(64, 64, 96, 128, 256)):
        super().__init__()
        self.convnormrelu = conv_norm_act(3, layers[0], norm, 7, 2)
        self.layer1 = nn.Sequential(ResidualBlock(layers[0], layers[1], norm, 1), ResidualBlock(layers[1], layers[1], norm, 1))
        self.layer2 = nn.Sequential(ResidualBlock(layers[1], layers[2], norm, 2), ResidualBlock(layers[2], layers[2], norm, 1))
        self.layer3 = nn.Sequential(ResidualBlock(layers[2], layers[3], norm, 2), ResidualBlock(layers[3], layers[3], norm, 1))
        self.conv = nn.Conv2d(layers[3], layers[4], 1)

    def forward(self, x):
        return self.conv(self.layer3(self.layer2(self.layer1(self.convnormrelu(x)))))


class MotionEncoder(nn.Module):
    def __init__(self, in_channels_corr, corr_layers=(256, 192), flow_layers=(128, 64), out_channels=128):
        super().__init__()
        self.convcorr1 = conv_norm_act(in_channels_corr, corr_layers[0], None, 1)
        self.convcorr2 = conv_norm_act(corr_layers[0], corr_layers[1], None, 3)
        self.convflow1 = conv_norm_act(2, flow_layers[0], None, 7)
        self.convflow2 = conv_norm_act(flow_layers[0], flow_layers[1], None, 3)
        self.conv = conv_norm_act(corr_layers[1] + flow_layers[1], out_channels - 2, None, 3)

    def forward(self, flow, corr_features):
        corr = self.convcorr2(self.convcorr1(corr_features))
        f = self.convflow2(self.convflow1(flow))
        return torch.cat([self.conv(torch.cat([corr, f], dim=1)), flow], dim=1)


class ConvGRU(nn.Module):
    def __init__(self, input_size, hidden_size, kernel_size, padding):
        super().__init__()
        self.convz = nn.Conv2d(hidden_size + input_size, hidden_size, kernel_size, padding=padding)
        self.convr = nn.Conv2d(hidden_size + input_size, hidden_size, kernel_size, padding=padding)
        self.convq = nn.Conv2d(hidden_size + input_size, hidden_size, kernel_size, padding=padding)

    def forward(self, h, x):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(self.convz(hx))
        r = torch.sigmoid(self.convr(hx))
        q = torch.tanh(self.convq(torch.cat([r * h, x], dim=1)))
        return (1 - z) * h + z * q


class RecurrentBlock(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.convgru1 = ConvGRU(input_size, hidden_size, (1, 5), (0, 2))
        self.convgru2 = ConvGRU(input_size, hidden_size, (5, 1), (2, 0))

    def forward(self, h, x):
        return self.convgru2(self.convgru1(h, x), x)


class FlowHead(nn.Module):
    def __init__(self, in_channels, hidden_size):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, hidden_size, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_size, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class UpdateBlock(nn.Module):
    def __init__(self, corr_channels, hidden=128, context=128):
        super().__init__()
        self.motion_encoder = MotionEncoder(corr_channels)
        self.recurrent_block = RecurrentBlock(128 + context, hidden)
        self.flow_head = FlowHead(hidden, 256)

    def forward(self, hidden_state, context, corr_features, flow):
        x = torch.cat([context, self.motion_encoder(flow, corr_features)], dim=1)
        hidden_state = self.recurrent_block(hidden_state, x)
        return hidden_state, self.flow_head(hidden_state)


class MaskPredictor(nn.Module):
    def __init__(self, in_channels=128, hidden_size=256, multiplier=0.25):
        super().__init__()
        self.convrelu = conv_norm_act(in_channels, hidden_size, None, 3)
        self.conv = nn.Conv2d(hidden_size, 8 * 8 * 9, 1)
        self.multiplier = multiplier

    def forward(self, x):
        return self.multiplier * self.conv(self.convrelu(x))


def grid_sample_abs(img, grid):
    """torchvision.models.optical_flow._utils.grid_sample: absolute pixel coordinates, bilinear, align_corners=True, zero padding."""
    h, w = img.shape[-2:]
    x, y = grid.split([1, 1], dim=-1)
    x = 2 * x / (w - 1) - 1
    if h > 1:
        y = 2 * y / (h - 1) - 1
    return F.grid_sample(img, torch.cat([x, y], dim=-1), mode="bilinear", align_corners=True)


def coords_grid(b, h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(b, 1, 1, 1)    # channel 0 = x, 1 = y


class CorrBlock:
    def __init__(self, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.out_channels = num_levels * (2 * radius + 1) ** 2

    def build_pyramid(self, fmap1, fmap2):
        b, c, h, w = fmap1.shape
        corr = torch.matmul(fmap1.view(b, c, h * w).transpose(1, 2), fmap2.view(b, c, h * w)) / torch.sqrt(torch.tensor(float(c)))
        corr = corr.reshape(b * h * w, 1, h, w)
        self.pyramid = [corr]
        for _ in range(self.num_levels - 1):
            corr = F.avg_pool2d(corr, kernel_size=2, stride=2)
            self.pyramid.append(corr)

    def index_pyramid(self, centroids):
        side = 2 * self.radius + 1
        d = torch.linspace(-self.radius, self.radius, side)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, side, side, 2)   # delta[i][j] = (d_i -> x, d_j -> y)
        b, _, h, w = centroids.shape
        c = centroids.permute(0, 2, 3, 1).reshape(b * h * w, 1, 1, 2)
        out = []
        for corr in self.pyramid:
            out.append(grid_sample_abs(corr, c + delta).view(b, h, w, -1))
            c = c / 2
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


def upsample_flow(flow, up_mask, factor=8):
    b, c, h, w = flow.shape
    up_mask = torch.softmax(up_mask.view(b, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, kernel_size=3, padding=1).view(b, c, 9, 1, 1, h, w)
    up = torch.sum(up_mask * up, dim=2)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(b, c, h * factor, w * factor)


class RAFT(nn.Module):
    """torchvision raft_large (state-dict keys feature_encoder.*, context_encoder.*, update_block.*, mask_predictor.*)."""

    def __init__(self):
        super().__init__()
        self.feature_encoder = FeatureEncoder(nn.InstanceNorm2d)
        self.context_encoder = FeatureEncoder(nn.BatchNorm2d)
        self.corr_block = CorrBlock(4, 4)
        self.update_block = UpdateBlock(self.corr_block.out_channels)
        self.mask_predictor = MaskPredictor()

    def forward(self, image1, image2, num_flow_updates=12):
        b, _, h, w = image1.shape
        if image2.shape[-2:] != (h, w) or h % 8 or w % 8:
            raise ValueError("input images must share a shape divisible by 8")
        fmap1, fmap2 = torch.chunk(self.feature_encoder(torch.cat([image1, image2], dim=0)), 2, dim=0)
        self.corr_block.build_pyramid(fmap1, fmap2)
        hidden_state, context = torch.split(self.context_encoder(image1), [128, 128], dim=1)
        hidden_state, context = torch.tanh(hidden_state), F.relu(context)
        coords0, coords1 = coords_grid(b, h // 8, w // 8), coords_grid(b, h // 8, w // 8)
        flows = []
        for _ in range(num_flow_updates):
            corr_features = self.corr_block.index_pyramid(coords1)
            hidden_state, delta = self.update_block(hidden_state, context, corr_features, coords1 - coords0)
            coords1 = coords1 + delta
            flows.append(upsample_flow(coords1 - coords0, self.mask_predictor(hidden_state)))
        return flows


class RAFTFlow(nn.Module):
    """flow_utils.py:134-189 (``img_size`` resize branch included: antialias=False bilinear, then resize_flow back)."""

    def __init__(self):
        super().__init__()
        self.model = RAFT().eval()

    @torch.no_grad()
    def forward(self, img1, img2, img_size=None, num_flow_updates=12):
        from .flow import resize_flow
        original = img1.shape[2:]
        if img_size is not None:
            img1 = F.interpolate(img1, size=img_size, mode="bilinear", align_corners=False, antialias=False)
            img2 = F.interpolate(img2, size=img_size, mode="bilinear", align_corners=False, antialias=False)
        img1, img2 = (img1.float() - 0.5) / 0.5, (img2.float() - 0.5) / 0.5   # the preset's normalize(mean 0.5, std 0.5)
        flow = self.model(img1.contiguous(), img2.contiguous(), num_flow_updates)[-1]
        if img_size is not None:
            flow = resize_flow(flow, original)
        return flow
