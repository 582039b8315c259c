"""CLIP text encoder on the HIP kernels: drop-in for ``FrozenCLIPEmbedder`` (modules/openclip/modules.py:88-135),
SURVEY.md 8(f) row 1.

``FrozenCLIPEmbedder(...).encode(list[str]) -> [n, 77, 768]`` exactly as the reference: tokenise (max_length 77, padded),
run the CLIP ViT-L/14 text tower, return ``last_hidden_state`` / ``pooler_output[:, None]`` / ``hidden_states[layer_idx]``.
The transformer (transformers ``CLIPTextModel``: token + position embedding, 12 pre-LN blocks of causal self-attention
and quick-GELU MLP, final LayerNorm) runs entirely in libinsv2v_hip.so: ``insv2v_embed_tokens``, LayerNorm folded into the
QKV / fc1 GEMMs, ``insv2v_attention`` with the causal flag, fused bias / residual / quick-GELU epilogues.

Tokenisation is host string processing and needs the CLIP BPE vocabulary (``vocab.json`` + ``merges.txt``), which is not
available offline: pass ``tokenizer=`` (any callable with the transformers tokenizer call signature), or a local directory
as ``version`` (loaded with ``transformers.CLIPTokenizer``), or feed token ids to ``encode_ids``.
"""
import torch

from . import ops
from .unet import fold_layernorm, _dev, prep_linear, prep_norm

_PREFIXES = ("transformer.", "text_model.")


def _strip(sd):
    out = {}
    for k, v in sd.items():
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        out[k] = v
    out.pop("embeddings.position_ids", None)  # modules.py:133
    return out


class CLIPTextTransformer:
    """transformers CLIPTextModel forward (causal, quick_gelu) on device; weights fp16, statistics / accumulation fp32."""

    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, device="cuda", **unused):
        if hidden_size % num_attention_heads or (hidden_size // num_attention_heads) not in (16, 32, 64, 128):
            raise ValueError("head_dim must be one of 16/32/64/128")
        self.cfg = dict(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                        num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                        max_position_embeddings=max_position_embeddings)
        self.eps, self.device, self.layers = layer_norm_eps, torch.device(device), None

    def load_state_dict(self, state_dict, strict=True):
        sd, dev, c = _strip(state_dict), self.device, self.cfg
        self.tok = _dev(sd["embeddings.token_embedding.weight"], torch.float16, dev)
        self.pos = _dev(sd["embeddings.position_embedding.weight"], torch.float16, dev)
        if tuple(self.tok.shape) != (c["vocab_size"], c["hidden_size"]):
            raise RuntimeError(f"token_embedding is {tuple(self.tok.shape)}, config says {(c['vocab_size'], c['hidden_size'])}")
        self.layers = []
        for i in range(c["num_hidden_layers"]):
            p = f"encoder.layers.{i}."
            wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"].float() for n in "qkv"], 0)
            bqkv = torch.cat([sd[p + f"self_attn.{n}_proj.bias"].float() for n in "qkv"], 0)
            wf, cs, b = fold_layernorm(wqkv, sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], bqkv)
            w1, cs1, b1 = fold_layernorm(sd[p + "mlp.fc1.weight"], sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"],
                                         sd[p + "mlp.fc1.bias"])
            f32 = torch.float32
            self.layers.append(dict(
                wqkv=_dev(wf, torch.float16, dev), qkv_cs=_dev(cs, f32, dev), qkv_b=_dev(b, f32, dev),
                wo=prep_linear(sd, p + "self_attn.out_proj", dev),
                w1=_dev(w1, torch.float16, dev), cs1=_dev(cs1, f32, dev), b1=_dev(b1, f32, dev),
                w2=prep_linear(sd, p + "mlp.fc2", dev)))
        self.final_ln = prep_norm(sd, "final_layer_norm", dev)
        return self

    def __call__(self, input_ids, output_hidden_states=False):
        if self.layers is None:
            raise RuntimeError("CLIPTextTransformer: load_state_dict() first")
        c = self.cfg
        ids = torch.as_tensor(input_ids, dtype=torch.long)
        if ids.dim() != 2:
            raise ValueError("input_ids must be [n, L]")
        if int(ids.min()) < 0 or int(ids.max()) >= c["vocab_size"]:
            raise IndexError("token id out of range")  # nn.Embedding raises for the reference
        n, L = ids.shape
        C, H = c["hidden_size"], c["num_attention_heads"]
        d = C // H
        x = ops.embed_tokens(ids.to(self.device).contiguous(), self.tok, self.pos)  # raises ValueError if L > max positions
        hidden = [x] if output_hidden_states else None
        for ly in self.layers:
            qkv = ops.gemm(x, ly["wqkv"], ly["qkv_b"], row_stats=ops.layernorm_stats(x, self.eps), col_sum=ly["qkv_cs"])
            a = torch.empty((n * L, C), device=x.device, dtype=torch.float16)
            p = qkv.data_ptr()
            addr = (1, L * 3 * C, 0)
            ops.attention(p, p + 2 * C, p + 4 * C, a, batch=n, heads=H, head_dim=d, seq_q=L, seq_k=L, scale=d ** -0.5,
                          q_rs=3 * C, k_rs=3 * C, v_rs=3 * C, o_rs=C, q_addr=addr, kv_addr=addr, o_addr=(1, L * C, 0), causal=True)
            x = ops.gemm(a, *ly["wo"], residual=x)
            h = ops.gemm(x, ly["w1"], ly["b1"], act=ops.ACT_QUICK_GELU, row_stats=ops.layernorm_stats(x, self.eps), col_sum=ly["cs1"])
            x = ops.gemm(h, *ly["w2"], residual=x)
            if output_hidden_states:
                hidden.append(x)
        last = ops.layernorm(x, *self.final_ln, eps=self.eps).reshape(n, L, C)
        eot = ids.argmax(dim=-1).to(self.device)  # legacy eos_token_id == 2 pooling of openai/clip-vit-large-patch14
        out = dict(last_hidden_state=last, pooler_output=last[torch.arange(n, device=self.device), eot])
        if output_hidden_states:
            out["hidden_states"] = [h.reshape(n, L, C) for h in hidden]
        return out


class FrozenCLIPEmbedder:
    """Same constructor / ``forward`` / ``encode`` / ``load_state_dict`` surface as modules/openclip/modules.py:88-135."""
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, tokenizer=None, config=None):
        assert layer in self.LAYERS
        if layer == "hidden":
            assert layer_idx is not None
            assert 0 <= abs(layer_idx) <= 12
        self.device, self.max_length, self.layer, self.layer_idx = device, max_length, layer, layer_idx
        self.transformer = CLIPTextTransformer(device=device, **(config or {}))
        self.tokenizer = tokenizer if tokenizer is not None else self._local_tokenizer(version)

    @staticmethod
    def _local_tokenizer(version):
        try:  # only a local directory can work offline; never reaches for the network
            from transformers import CLIPTokenizer
            return CLIPTokenizer.from_pretrained(version, local_files_only=True)
        except Exception:
            return None

    def freeze(self):
        return self  # inference-only implementation: nothing requires grad

    def load_state_dict(self, state_dict, strict=True):
        self.transformer.load_state_dict(state_dict, strict)
        return self

    def encode_ids(self, tokens):
        out = self.transformer(tokens, output_hidden_states=self.layer == "hidden")
        if self.layer == "last":
            z = out["last_hidden_state"]
        elif self.layer == "pooled":
            z = out["pooler_output"][:, None, :]
        else:
            z = out["hidden_states"][self.layer_idx]
        return z.float()  # the reference returns fp32 embeddings

    def forward(self, text):
        if self.tokenizer is None:
            raise RuntimeError("FrozenCLIPEmbedder: no CLIP tokenizer available offline (vocab.json / merges.txt); pass "
                               "tokenizer=, a local directory as version=, or call encode_ids(token_ids)")
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return self.encode_ids(enc["input_ids"])

    __call__ = forward

    def encode(self, text):
        return self(text)
