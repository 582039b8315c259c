"""State-dict key -> shape enumeration for the UNet and VAE (reference key names),
used to create random-init weights of the real architecture without instantiating torch modules."""


def _lin(d, k, o, i, bias=True):
    d[k + ".weight"] = (o, i)
    if bias:
        d[k + ".bias"] = (o,)


def _conv(d, k, o, i, ks):
    d[k + ".weight"] = (o, i, ks, ks)
    d[k + ".bias"] = (o,)


def _norm(d, k, c):
    d[k + ".weight"] = (c,)
    d[k + ".bias"] = (c,)


def _attn(d, k, dim, ctx=None):
    for n in "qkv":
        _lin(d, f"{k}.to_{n}", dim, dim if (ctx is None or n == "q") else ctx, bias=False)
    _lin(d, f"{k}.to_out.0", dim, dim)


def _ff(d, k, dim):
    _lin(d, f"{k}.net.0.proj", dim * 8, dim)
    _lin(d, f"{k}.net.2", dim, dim * 4)


def _res(d, k, cin, cout, temb):
    _norm(d, k + ".norm1", cin)
    _conv(d, k + ".conv1", cout, cin, 3)
    _lin(d, k + ".time_emb_proj", cout, temb)
    _norm(d, k + ".norm2", cout)
    _conv(d, k + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, k + ".conv_shortcut", cout, cin, 1)


def _spatial(d, k, ch, ctx):
    _norm(d, k + ".norm", ch)
    _conv(d, k + ".proj_in", ch, ch, 1)
    b = k + ".transformer_blocks.0"
    _attn(d, b + ".attn1", ch)
    _norm(d, b + ".norm1", ch)
    _attn(d, b + ".attn2", ch, ctx)
    _norm(d, b + ".norm2", ch)
    _ff(d, b + ".ff", ch)
    _norm(d, b + ".norm3", ch)
    _conv(d, k + ".proj_out", ch, ch, 1)


def _motion(d, k, ch, mkw):
    k = k + ".temporal_transformer"
    _norm(d, k + ".norm", ch)
    _lin(d, k + ".proj_in", ch, ch)
    for bi in range(mkw.get("num_transformer_block", 2)):
        b = f"{k}.transformer_blocks.{bi}"
        for ai in range(len(mkw.get("attention_block_types", ("Temporal_Self", "Temporal_Self")))):
            _attn(d, f"{b}.attention_blocks.{ai}", ch)
            d[f"{b}.attention_blocks.{ai}.pos_encoder.pe"] = (1, mkw.get("temporal_position_encoding_max_len", 24), ch)
            _norm(d, f"{b}.norms.{ai}", ch)
        _ff(d, b + ".ff", ch)
        _norm(d, b + ".ff_norm", ch)
    _lin(d, k + ".proj_out", ch, ch)


def unet_shapes(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                down_block_types=("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",),
                up_block_types=("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3, layers_per_block=2,
                cross_attention_dim=1280, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
                motion_module_mid_block=True, motion_module_decoder_only=False, motion_module_kwargs=None, **unused):
    d, ch, mkw = {}, list(block_out_channels), dict(motion_module_kwargs or {})
    temb = ch[0] * 4
    _conv(d, "conv_in", ch[0], in_channels, 3)
    _lin(d, "time_embedding.linear_1", temb, ch[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    out = ch[0]
    for i, typ in enumerate(down_block_types):
        cin, out = out, ch[i]
        mot = use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only
        for j in range(layers_per_block):
            _res(d, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
            if typ.startswith("CrossAttn"):
                _spatial(d, f"down_blocks.{i}.attentions.{j}", out, cross_attention_dim)
            if mot:
                _motion(d, f"down_blocks.{i}.motion_modules.{j}", out, mkw)
        if i != len(ch) - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    _res(d, "mid_block.resnets.0", ch[-1], ch[-1], temb)
    _res(d, "mid_block.resnets.1", ch[-1], ch[-1], temb)
    _spatial(d, "mid_block.attentions.0", ch[-1], cross_attention_dim)
    if use_motion_module and motion_module_mid_block:
        _motion(d, "mid_block.motion_modules.0", ch[-1], mkw)
    rev = ch[::-1]
    out = rev[0]
    for i, typ in enumerate(up_block_types):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, len(ch) - 1)]
        mot = use_motion_module and (2 ** (3 - i) in motion_module_resolutions)
        n = layers_per_block + 1
        for j in range(n):
            _res(d, f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out) + (cin if j == n - 1 else out), out, temb)
            if typ.startswith("CrossAttn"):
                _spatial(d, f"up_blocks.{i}.attentions.{j}", out, cross_attention_dim)
            if mot:
                _motion(d, f"up_blocks.{i}.motion_modules.{j}", out, mkw)
        if i != len(ch) - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(d, "conv_norm_out", ch[0])
    _conv(d, "conv_out", out_channels, ch[0], 3)
    return d


def _vres(d, k, cin, cout):
    _norm(d, k + ".norm1", cin)
    _conv(d, k + ".conv1", cout, cin, 3)
    _norm(d, k + ".norm2", cout)
    _conv(d, k + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, k + ".nin_shortcut", cout, cin, 1)


def _vattn(d, k, ch):
    _norm(d, k + ".norm", ch)
    for n in ("q", "k", "v", "proj_out"):
        _conv(d, f"{k}.{n}", ch, ch, 1)


def vae_shapes(ddconfig, embed_dim=4, **unused):
    dd = ddconfig
    ch, mult, nres, z = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"], dd["z_channels"]
    d = {}
    _conv(d, "encoder.conv_in", ch, dd["in_channels"], 3)
    cur, inm = ch, [1] + mult
    for lvl in range(len(mult)):
        cur = ch * inm[lvl]
        for j in range(nres):
            _vres(d, f"encoder.down.{lvl}.block.{j}", cur, ch * mult[lvl])
            cur = ch * mult[lvl]
        if lvl != len(mult) - 1:
            _conv(d, f"encoder.down.{lvl}.downsample.conv", cur, cur, 3)
    _vres(d, "encoder.mid.block_1", cur, cur)
    _vattn(d, "encoder.mid.attn_1", cur)
    _vres(d, "encoder.mid.block_2", cur, cur)
    _norm(d, "encoder.norm_out", cur)
    _conv(d, "encoder.conv_out", 2 * z if dd.get("double_z", True) else z, cur, 3)
    cur = ch * mult[-1]
    _conv(d, "decoder.conv_in", cur, z, 3)
    _vres(d, "decoder.mid.block_1", cur, cur)
    _vattn(d, "decoder.mid.attn_1", cur)
    _vres(d, "decoder.mid.block_2", cur, cur)
    for lvl in reversed(range(len(mult))):
        for j in range(nres + 1):
            _vres(d, f"decoder.up.{lvl}.block.{j}", cur, ch * mult[lvl])
            cur = ch * mult[lvl]
        if lvl != 0:
            _conv(d, f"decoder.up.{lvl}.upsample.conv", cur, cur, 3)
    _norm(d, "decoder.norm_out", cur)
    _conv(d, "decoder.conv_out", dd["out_ch"], cur, 3)
    _conv(d, "quant_conv", 2 * embed_dim, 2 * z, 1)
    _conv(d, "post_quant_conv", z, embed_dim, 1)
    return d


def clip_text_shapes(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                     max_position_embeddings=77, prefix="transformer.text_model.", **unused):
    """FrozenCLIPEmbedder state dict as stored in the reference checkpoint (``text_model.`` + these keys)."""
    d = {}
    C = hidden_size
    d[prefix + "embeddings.token_embedding.weight"] = (vocab_size, C)
    d[prefix + "embeddings.position_embedding.weight"] = (max_position_embeddings, C)
    for i in range(num_hidden_layers):
        k = f"{prefix}encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(d, f"{k}.self_attn.{n}", C, C)
        _norm(d, f"{k}.layer_norm1", C)
        _lin(d, f"{k}.mlp.fc1", intermediate_size, C)
        _lin(d, f"{k}.mlp.fc2", C, intermediate_size)
        _norm(d, f"{k}.layer_norm2", C)
    _norm(d, prefix + "final_layer_norm", C)
    return d


def raft_shapes(prefix=""):
    """torchvision ``raft_large`` state dict (flow_utils.py:157-158 loads ``Raft_Large_Weights.DEFAULT`` into it): key -> shape.
    InstanceNorm2d (feature encoder) has no parameters; BatchNorm2d (context encoder) carries weight / bias / running statistics."""
    d = {}

    def conv(k, o, i, kh, kw=None):
        d[prefix + k + ".weight"] = (o, i, kh, kh if kw is None else kw)
        d[prefix + k + ".bias"] = (o,)

    def bn(k, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            d[prefix + f"{k}.{n}"] = (c,)
        d[prefix + k + ".num_batches_tracked"] = ()

    for enc, batch in (("feature_encoder", False), ("context_encoder", True)):
        def cnr(k, o, i, ks):
            conv(f"{enc}.{k}.0", o, i, ks)
            if batch:
                bn(f"{enc}.{k}.1", o)
        cnr("convnormrelu", 64, 3, 7)
        cin = 64
        for li, cout in ((1, 64), (2, 96), (3, 128)):
            for bi in (0, 1):
                i = cin if bi == 0 else cout
                cnr(f"layer{li}.{bi}.convnormrelu1", cout, i, 3)
                cnr(f"layer{li}.{bi}.convnormrelu2", cout, cout, 3)
                if bi == 0 and li > 1:
                    cnr(f"layer{li}.{bi}.downsample", cout, i, 1)
            cin = cout
        conv(f"{enc}.conv", 256, 128, 1)
    m = "update_block.motion_encoder"
    conv(m + ".convcorr1.0", 256, 324, 1)
    conv(m + ".convcorr2.0", 192, 256, 3)
    conv(m + ".convflow1.0", 128, 2, 7)
    conv(m + ".convflow2.0", 64, 128, 3)
    conv(m + ".conv.0", 126, 256, 3)
    for name, (kh, kw) in (("convgru1", (1, 5)), ("convgru2", (5, 1))):
        for g in ("convz", "convr", "convq"):
            conv(f"update_block.recurrent_block.{name}.{g}", 128, 384, kh, kw)
    conv("update_block.flow_head.conv1", 256, 128, 3)
    conv("update_block.flow_head.conv2", 2, 256, 3)
    conv("mask_predictor.convrelu.0", 256, 128, 3)
    conv("mask_predictor.conv", 576, 256, 1)
    return d
