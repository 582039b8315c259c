"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the CLIP text encoder behind the reference's
``FrozenCLIPEmbedder`` (modules/openclip/modules.py:88-135).

The arithmetic lives in the third-party ``transformers`` package (``CLIPTextModel``; the reference's
requirements.txt:4 leaves the version unpinned), so it is restated here from the published algorithm
(``transformers/models/clip/modeling_clip.py``: CLIPTextEmbeddings, CLIPAttention, CLIPMLP with
``hidden_act="quick_gelu"``, CLIPEncoderLayer pre-LN residual blocks, causal mask, ``final_layer_norm``).
PINNED: ``transformers`` (5.15.0) IS installed in the build container and on the GPU box, so
``tools/gen_golden.py`` (case ``clip_text``) and ``tests/test_cpu_oracle.py`` run the real ``CLIPTextModel``
on the same key-hashed weights and assert equality with this restatement; the golden vectors under
``tests/golden/clip_text_*.npz`` are the real model's outputs.

State-dict keys are the checkpoint's (``insv2v.pth``: ``text_model.transformer.text_model.*``) with the
``transformer.`` / ``text_model.`` prefixes optional.
"""
import torch
import torch.nn.functional as F

_PREFIXES = ("transformer.", "text_model.")


def strip_prefixes(sd):
    """Accept ``transformer.text_model.X`` (reference checkpoint), ``text_model.X`` (transformers 4.x) or ``X`` (5.x)."""
    out = {}
    for k, v in sd.items():
        changed = True
        while changed:
            changed = False
            for p in _PREFIXES:
                if k.startswith(p):
                    k, changed = k[len(p):], True
        if k != "embeddings.position_ids":  # dropped by newer transformers, popped at modules.py:133
            out[k] = v
    return out


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def clip_text_forward(sd, input_ids, num_heads, eps=1e-5):
    """-> dict(last_hidden_state [n,L,C], pooler_output [n,C], hidden_states list of L+1 tensors [n,L,C]).

    modules.py:114-125 selects ``last_hidden_state`` (layer="last"), ``pooler_output[:, None]`` ("pooled") or
    ``hidden_states[layer_idx]`` ("hidden")."""
    sd = {k: v.float() for k, v in strip_prefixes(sd).items()}
    n, L = input_ids.shape
    tok, pos = sd["embeddings.token_embedding.weight"], sd["embeddings.position_embedding.weight"]
    if L > pos.shape[0]:
        raise ValueError(f"Sequence length must be less than max_position_embeddings (got {L} > {pos.shape[0]})")
    x = tok[input_ids] + pos[:L][None]
    C = x.shape[-1]
    d = C // num_heads
    causal = torch.full((L, L), float("-inf")).triu(1)
    hidden = [x]
    nl = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(nl):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (C,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * d ** -0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.reshape(n, L, num_heads, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + causal, dim=-1) @ v
        a = a.transpose(1, 2).reshape(n, L, C)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (C,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = quick_gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        hidden.append(x)
    last = F.layer_norm(x, (C,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps)
    # openai/clip-vit-large-patch14 ships eos_token_id = 2 in its config -> the legacy argmax pooling (EOT has the highest id)
    pooled = last[torch.arange(n), input_ids.argmax(dim=-1)]
    return dict(last_hidden_state=last, pooler_output=pooled, hidden_states=hidden)


def embed(sd, input_ids, num_heads, layer="last", layer_idx=None):
    """FrozenCLIPEmbedder.forward after tokenisation (modules.py:118-125)."""
    out = clip_text_forward(sd, input_ids, num_heads)
    if layer == "last":
        return out["last_hidden_state"]
    if layer == "pooled":
        return out["pooler_output"][:, None, :]
    return out["hidden_states"][layer_idx]
