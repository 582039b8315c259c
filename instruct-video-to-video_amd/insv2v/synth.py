"""Deterministic synthetic weights and inputs (no checkpoints / datasets offline).

Every tensor is drawn from a CPU generator seeded with crc32(key), so the same
state dict is reproduced on any machine without shipping weight fixtures
(SURVEY.md §8d).  Motion-module ``proj_out`` (zero-initialised in the reference,
motion_module.py:68-69) is made NON-zero so temporal attention is observable.
"""
import zlib
import torch


def _gen(key, salt=0):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def _pe_table(shape):
    """motion_module.py:229-233 sinusoid buffer [1, max_len, d_model]."""
    import math
    _, max_len, d_model = shape
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(pos * div)
    pe[0, :, 1::2] = torch.cos(pos * div)
    return pe


def synth_tensor(key, ref, salt=0):
    """Synthetic value for state-dict entry ``key``; ``ref`` is a tensor or a shape tuple."""
    g = _gen(key, salt)
    shape = tuple(ref.shape) if hasattr(ref, "shape") else tuple(ref)
    if key.endswith("pos_encoder.pe"):
        return ref.clone().float() if hasattr(ref, "clone") else _pe_table(shape)
    is_norm = any(s in key for s in (".norm", "norm1.", "norm2.", "norm3.", "norms.", "ff_norm", "norm_out", "conv_norm_out", "layer_norm")) \
        or key.startswith("norm")
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_var"):            # BatchNorm statistics (the optical-flow context encoder): positive
        return 0.5 + torch.rand(shape, generator=g)
    if len(shape) == 1:
        r = torch.randn(shape, generator=g)
        if is_norm and key.endswith("weight"):
            return 1.0 + 0.1 * r
        return 0.1 * r
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (fan_in ** -0.5)


def synth_state_dict(module_or_sd, salt=0, prefix=""):
    """module / state dict / {key: shape} dict -> synthetic state dict (keys hashed with ``prefix``)."""
    sd = module_or_sd if isinstance(module_or_sd, dict) else module_or_sd.state_dict()
    return {k: synth_tensor(prefix + k, v, salt) for k, v in sd.items()}


def synth_raft_state_dict(shapes_dict, salt=0):
    """Key-hashed weights of the optical-flow estimator (shapes.raft_shapes()): plain fan-in scaling.  (Probed on the CPU oracle: with He
    gain the 12 recurrent updates amplify a 1e-3 input perturbation to 2.4e-2 of the flow - useless for an fp16-vs-fp32 comparison; with
    fan-in scaling the same perturbation stays at 3.6e-4 while the correspondences still move ~2 px per update.)"""
    return {k: synth_tensor("raft." + k, shp, salt) for k, shp in shapes_dict.items()}


def synth_input(name, shape, kind="normal", scale=1.0, salt=0):
    g = _gen("input:" + name, salt)
    if kind == "uniform":
        return (torch.rand(shape, generator=g) * 2 - 1) * scale
    return torch.randn(shape, generator=g) * scale


UNET_FULL = dict(
    in_channels=8, out_channels=4, act_fn="silu", attention_head_dim=8,
    block_out_channels=[320, 640, 1280, 1280], cross_attention_dim=768,
    down_block_types=["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"],
    up_block_types=["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3,
    downsample_padding=1, layers_per_block=2, mid_block_scale_factor=1, norm_eps=1e-5,
    norm_num_groups=32, sample_size=64, use_motion_module=True,
    motion_module_resolutions=[1, 2, 4, 8], motion_module_mid_block=False,
    motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                              temporal_attention_dim_div=1),
)

# Same wiring, reduced width: channels are multiples of 64 so the
# HIP implicit-GEMM K-slices never straddle a 3x3 tap.
UNET_TINY = dict(UNET_FULL, block_out_channels=[64, 128, 256, 256], attention_head_dim=4,
                 cross_attention_dim=64,
                 motion_module_kwargs=dict(UNET_FULL["motion_module_kwargs"], num_attention_heads=4))

VAE_FULL = dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3,
                                           out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
                                           attn_resolutions=[], dropout=0.0))
VAE_TINY = dict(embed_dim=4, ddconfig=dict(VAE_FULL["ddconfig"], ch=64, ch_mult=[1, 2, 2, 2], num_res_blocks=1))

# CLIP ViT-L/14 text tower (openai/clip-vit-large-patch14, modules/openclip/modules.py:96) and a reduced-width variant
CLIP_FULL = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77)
CLIP_TINY = dict(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                 max_position_embeddings=77)


def synth_token_ids(name, n, L, vocab, salt=0):
    """Token ids shaped like CLIP tokeniser output: BOS, words, EOS (= highest id), EOS padding."""
    g = _gen("ids:" + name, salt)
    ids = torch.full((n, L), vocab - 1, dtype=torch.long)
    ids[:, 0] = vocab - 2
    for i in range(n):
        m = int(torch.randint(3, L - 2, (1,), generator=g))
        ids[i, 1:1 + m] = torch.randint(0, vocab - 2, (m,), generator=g)
    return ids
