"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, argument
structs match the header, shape enumeration, weight preparation, config loading, and the N>1
clip-parallel path on gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, PKG


def header_text():
    return open(os.path.join(ROOT, "include", "insv2v_hip.h")).read()


def test_library_loads_and_exports_every_declared_symbol():
    from insv2v import _lib
    if not os.path.exists(_lib.LIB_PATH):
        sys.path.insert(0, PKG)
        import build
        build.build(verbose=False)
    lib = _lib.load()
    declared = set(re.findall(r"^\s*int\s+(insv2v_\w+)\s*\(", header_text(), flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.insv2v_abi_version() == _lib.ABI_VERSION


def _c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header_text(), flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const\s+)?(void|float|int64_t|int32_t)\s*(\*?)\s*(.*)", decl)
        ctype = ("ptr" if m.group(3) or "*" in m.group(4) else m.group(2))
        for var in m.group(4).split(","):
            fields.append((var.strip().lstrip("*").strip(), ctype))
    return fields


@pytest.mark.parametrize("cname,pyname", [("insv2v_gemm_desc", "GemmDesc"), ("insv2v_groupnorm_desc", "GroupNormDesc"),
                                           ("insv2v_layernorm_desc", "LayerNormDesc"), ("insv2v_attention_desc", "AttentionDesc"),
                                           ("insv2v_step_desc", "StepDesc")])
def test_ctypes_structs_mirror_the_header(cname, pyname):
    from insv2v import _lib
    want = _c_struct_fields(cname)
    kind = {ctypes.c_void_p: "ptr", ctypes.c_int64: "int64_t", ctypes.c_int32: "int32_t", ctypes.c_float: "float"}
    got = [(n, kind[t]) for n, t in getattr(_lib, pyname)._fields_]
    assert got == want


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from insv2v import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipKernelError, match="no CPU fallback"):
        _lib.load()


def test_ops_reject_cpu_tensors():
    from insv2v import ops, _lib
    with pytest.raises(_lib.HipKernelError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16))
    with pytest.raises(_lib.HipKernelError):
        ops.layernorm(torch.zeros(8, 8, dtype=torch.float16), torch.ones(8), torch.zeros(8))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(PKG, "insv2v")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_shapes_match_oracle_state_dicts():
    import oracle.unet3d as ou, oracle.vae as ov
    from insv2v import shapes, synth
    for cfg in (synth.UNET_TINY, synth.UNET_FULL):
        sd = ou.UNet3DConditionModel(**cfg).state_dict()
        sh = shapes.unet_shapes(**cfg)
        assert set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sd)
    assert len(shapes.unet_shapes(**synth.UNET_FULL)) == 1246
    for cfg in (synth.VAE_TINY, synth.VAE_FULL):
        sd = ov.AutoencoderKL(**cfg).state_dict()
        sh = shapes.vae_shapes(**cfg)
        assert set(sd) == set(sh) and all(tuple(sd[k].shape) == tuple(sh[k]) for k in sd)


def test_weight_preparation_layouts():
    from insv2v.unet import prep_conv3x3, interleave32, sinusoid_table
    import oracle.unet3d as ou
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    wk, b = prep_conv3x3({"c.weight": w, "c.bias": torch.zeros(2)}, "c", "cpu")
    assert wk.shape == (2, 9 * 64)
    wk = wk.reshape(2, 3, 3, 64)
    assert torch.equal(wk[..., :3].float(), w.permute(0, 2, 3, 1)) and wk[..., 3:].abs().max() == 0
    t = torch.arange(128).float()[:, None].repeat(1, 2)  # h = rows 0..63, g = rows 64..127
    i = interleave32(t)[:, 0]
    assert i[:32].tolist() == list(range(32)) and i[32:64].tolist() == list(range(64, 96)) and i[64:96].tolist() == list(range(32, 64))
    assert torch.allclose(sinusoid_table(64, 32), ou.PosEnc(64, 32).pe[0])


def test_config_loader_coerces_numeric_strings(tmp_path):
    from insv2v.model import load_config
    p = tmp_path / "c.yaml"
    p.write_text("unet:\n  params:\n    norm_eps: 1e-05\n    act_fn: silu\n    block_out_channels:\n      - 64\n")
    c = load_config(str(p))
    assert c["unet"]["params"]["norm_eps"] == 1e-5 and c["unet"]["params"]["act_fn"] == "silu"


def test_pipeline_constructor_contract():
    from insv2v.inference import InferenceIP2PVideo

    class FakeUNet:
        device = torch.device("cpu")
    p = InferenceIP2PVideo(FakeUNet(), scheduler="ddpm", num_ddim_steps=20)
    assert p.scheduler.timesteps.tolist() == list(range(950, -1, -50))
    with pytest.raises(NotImplementedError):
        InferenceIP2PVideo(FakeUNet(), scheduler="pndm")


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from insv2v.clip_parallel import shard_units, gather_frames
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_units = 5
mine = shard_units(n_units, rank, world)
local = torch.stack([torch.full((2, 3), float(i)) for i in mine]) if mine else torch.zeros((0, 2, 3))
out = gather_frames(local, n_units)
assert out.shape == (n_units, 2, 3), out.shape
assert [int(out[i, 0, 0]) for i in range(n_units)] == list(range(n_units)), out[:, 0, 0]
dist.destroy_process_group()
print("ok", rank)
'''


def test_clip_parallel_gloo_world2(tmp_path):
    from insv2v.clip_parallel import shard_units, units_per_rank
    assert shard_units(16, 3, 8) == [3, 11] and units_per_rank(16, 8) == 2 and shard_units(5, 1, 2) == [1, 3]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), PKG]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


# ------------------------------------------------------------------------------------------- video I/O
def _make_dataset(root, n_frames=6, size=(40, 56)):
    from PIL import Image
    import numpy as np
    rows = [["Video name", "Original", "Style", "Object", "Background", "Multiple"],
            ["DAVIS Videos:", "", "", "", "", ""],
            ["gold-fish", "fish swim", "fish swim, watercolor", "sharks swim", "fish swim in space", "sharks swim in space, watercolor"],
            ["", "", "", "", "", ""],
            ["Youtube Videos:", "", "", "", "", ""],
            ["cat-walk", "a cat walks", "a cat walks, anime", "a dog walks", "a cat walks on the moon", "a dog walks on the moon, anime"]]
    import csv
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "LOVEU-TGVE-2023_Dataset.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    for folder, name in (("DAVIS_480p/480p_videos", "gold-fish"), ("youtube_480p/480p_videos", "cat-walk")):
        d = os.path.join(root, folder, name)
        os.makedirs(d, exist_ok=True)
        for i in range(n_frames):
            a = np.zeros((size[0], size[1], 3), dtype=np.uint8)
            a[..., 0], a[..., 1], a[: size[0] // 2, :, 2] = 10 * i, 200, 255
            Image.fromarray(a).save(os.path.join(d, f"{i:05d}.png"))
        open(os.path.join(d, "fps.txt"), "w").write("24")


def test_loveu_dataset_reader_and_writers(tmp_path):
    from PIL import Image
    from insv2v.video_io import LoveuTgveVideoDataset, save_tensor_to_gif, save_tensor_to_images, output_paths
    root = str(tmp_path / "loveu")
    _make_dataset(root)
    ds = LoveuTgveVideoDataset(root, image_size=(32, 24))  # (width, height) as cv2.resize takes it
    assert len(ds) == 2 and list(ds.data) == ["gold-fish", "cat-walk"]
    assert ds.data["gold-fish"]["source_folder"] == "DAVIS_480p/480p_videos"
    assert ds.data["cat-walk"]["source_folder"] == "youtube_480p/480p_videos"
    item = ds[1]
    assert item["video_name"] == "cat-walk" and item["object"] == "a dog walks" and item["fps"] == 24.0
    fr = item["frames"]
    assert fr.shape == (6, 3, 24, 32) and fr.dtype == torch.float32
    assert fr.min() >= -1 and fr.max() <= 1
    assert abs(fr[3, 0].mean().item() - (30 / 255 * 2 - 1)) < 1e-6 and abs(fr[0, 1].mean().item() - (200 / 255 * 2 - 1)) < 1e-6
    assert torch.equal(ds["gold-fish"]["frames"], ds[0]["frames"])
    # writers: [1,T,3,H,W] in [-1,1] -> GIF with T frames / T numbered JPGs
    gif, img_dir = output_paths("edit", 384, 3, 1.8, 7.5, 32, "cat-walk", "style", "a cat walks, anime")
    assert gif == "v2v_results/edit_prompt/loveu_tgve_384/gif/VID_3/VIDEO_CFG_1.8_TEXT_CFG_7.5/style_32_a_cat_walks,_anime.gif"
    assert img_dir == "v2v_results/edit_prompt/loveu_tgve_384/images_32/VIDEO_CFG_1.8_TEXT_CFG_7.5/cat-walk/style"
    out_gif = str(tmp_path / "o" / "x.gif")
    save_tensor_to_gif(fr[None], out_gif, fps=5)
    g = Image.open(out_gif)
    assert g.n_frames == 6 and g.size == (32, 24)
    save_tensor_to_images(fr[None], str(tmp_path / "imgs"))
    assert sorted(os.listdir(tmp_path / "imgs")) == [f"{i:03d}.jpg" for i in range(6)]
    assert Image.open(tmp_path / "imgs" / "000.jpg").size == (32, 24)
    with pytest.raises(FileNotFoundError):
        LoveuTgveVideoDataset(root).load_frames("missing", "DAVIS_480p/480p_videos")
