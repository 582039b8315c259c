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
    declared = set(re.findall(r"^\s*(?:int|int64_t)\s+(insv2v_\w+)\s*\(", header_text(), flags=re.M))
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
                                           ("insv2v_step_desc", "StepDesc"), ("insv2v_im2col_desc", "Im2colDesc"),
                                           ("insv2v_corr_lookup_desc", "CorrLookupDesc"), ("insv2v_winograd_in_desc", "WinogradInDesc"),
                                           ("insv2v_winograd_out_desc", "WinogradOutDesc")])
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
# unit counts: not divisible by world, fewer units than ranks (an EMPTY rank), one unit, exact multiple
for n_units in (5, 1, world - 1, 2 * world, 0 + world):
    if n_units <= 0:
        continue
    mine = shard_units(n_units, rank, world)
    item = (2, 3)
    # an empty rank only knows the item shape (as run_loveu_tgve.main does), either as a [0, ...] tensor or as None
    local = torch.stack([torch.full(item, float(i)) for i in mine]) if mine else (None if n_units % 2 else torch.zeros((0, *item)))
    out = gather_frames(local, n_units, item_shape=item, dtype=torch.float32)
    assert out.shape == (n_units, *item), (n_units, out.shape)
    assert [int(out[i, 0, 0]) for i in range(n_units)] == list(range(n_units)), out[:, 0, 0]
# the bug this guards against (run_loveu_tgve r1 built torch.zeros((0,)) on an empty rank): units whose shape
# contradicts item_shape are rejected on EVERY rank before the collective, instead of hanging in it
try:
    gather_frames(torch.zeros((1, 4)), world, item_shape=(2, 3))
    raise SystemExit("mismatched unit shape was accepted")
except ValueError:
    pass
dist.destroy_process_group()
print("ok", rank)
'''


@pytest.mark.parametrize("world", [2, 3])
def test_clip_parallel_gloo(tmp_path, world):
    from insv2v.clip_parallel import shard_units, units_per_rank
    assert shard_units(16, 3, 8) == [3, 11] and units_per_rank(16, 8) == 2 and shard_units(5, 1, 2) == [1, 3]
    assert shard_units(4, 6, 8) == [] and units_per_rank(4, 8) == 1  # 4 clips on 8 GPUs: ranks 4-7 own nothing
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + world), str(script), PKG]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == world


def test_optical_flow_cli_fails_fast_without_a_flow_source():
    # --with_optical_flow used to die with AttributeError after the first window had been sampled
    from insv2v.run_loveu_tgve import build_parser, check_args
    p = build_parser()
    # round 5: the RAFT estimator runs on the HIP kernels; its weights come from --raft-ckpt (or are key-hashed with --synthetic)
    with pytest.raises(SystemExit, match="--raft-ckpt"):
        check_args(p.parse_args(["--with_optical_flow"]))
    with pytest.raises(SystemExit, match="--units"):
        check_args(p.parse_args(["--with_optical_flow", "--flows", "f.pt"]))
    assert check_args(p.parse_args(["--with_optical_flow", "--flows", "f.pt", "--synthetic", "2"])).flows == "f.pt"
    assert check_args(p.parse_args(["--with_optical_flow", "--synthetic", "1"])).raft_ckpt is None
    assert check_args(p.parse_args(["--with_optical_flow", "--raft-ckpt", "raft.pth"])).raft_ckpt == "raft.pth"
    assert check_args(p.parse_args(["--synthetic", "1"])).flows is None
    from insv2v.run_loveu_tgve import optical_flow_pipe_kwargs
    assert optical_flow_pipe_kwargs(p.parse_args(["--synthetic", "1"])) == {}
    assert optical_flow_pipe_kwargs(p.parse_args(["--with_optical_flow", "--flows", "f.pt", "--synthetic", "2"])) == {}
    kw = optical_flow_pipe_kwargs(p.parse_args(["--with_optical_flow", "--synthetic", "1"]))
    assert set(kw) == {"raft_state_dict"} and "update_block.recurrent_block.convgru2.convq.weight" in kw["raft_state_dict"]


def test_edit_video_feeds_the_flow_variant_and_shares_the_encoded_video():
    # Host logic of edit_video with stand-in model / pipes (no GPU): (1) an optical-flow pipe without flows gets the
    # previous window's last R frames and the new frames (insv2v_run_loveu_tgve.py:141-160); (2) cond= skips the VAE encode
    from insv2v.run_loveu_tgve import edit_video

    class FakeModel:
        scale_factor = 0.5
        encodes = 0

        class unet:
            device = "cpu"

        def encode_image_to_latent(self, frames, noise=None):
            FakeModel.encodes += 1
            return torch.zeros(1, frames.shape[1], 4, 2, 2)

        def decode_latent_to_image(self, lat):
            return torch.zeros(1, lat.shape[1], 3, 16, 16)

    class FlowPipe:
        flow_estimator = staticmethod(lambda q, r: torch.zeros(len(r), 2, 16, 16))
        calls = []

        def obtain_flow_batched(self, *a):
            raise AssertionError

        def __call__(self, latent, **kw):
            return {"latent": latent}

        def second_clip_forward(self, latent, latent_ref, ref_images=None, query_images=None, flows=None, **kw):
            FlowPipe.calls.append((tuple(ref_images.shape), tuple(query_images.shape), flows))
            return {"latent": latent}

    frames = torch.arange(28.0).reshape(1, 28, 1, 1, 1).expand(1, 28, 3, 16, 16)
    out = edit_video(FakeModel(), FlowPipe(), frames, None, None)
    assert out.shape == (1, 28, 3, 16, 16) and FakeModel.encodes == 1
    assert FlowPipe.calls == [((1, 4, 3, 16, 16), (1, 12, 3, 16, 16), None)]
    edit_video(FakeModel(), FlowPipe(), frames, None, None, cond=torch.zeros(1, 28, 4, 2, 2))
    assert FakeModel.encodes == 1  # the shared posterior sample was used
    nof = FlowPipe()
    nof.flow_estimator = None
    with pytest.raises(RuntimeError, match="flow source"):
        edit_video(FakeModel(), nof, frames, None, None)


def test_runner_cache_is_shared_and_dropped_when_weights_are_reloaded(monkeypatch):
    # captured hipGraphs keep the old weight pointers; a reload must not replay them.  Graphs are shared
    # process-wide (pipes are short-lived, recapturing the same shape after destroying a graph crashed in hipGraphLaunch).
    import gc
    import insv2v.inference as inf

    class FakeUNet:
        device = "cpu"
        weights_version = 1

    made = []
    monkeypatch.setattr(inf, "GraphedUNet", lambda *a, **k: made.append(a) or object())
    monkeypatch.setattr(inf.torch.cuda, "synchronize", lambda: None)
    u = FakeUNet()
    p1 = inf.InferenceIP2PVideo(u, scheduler="ddim", num_ddim_steps=2)
    p2 = inf.InferenceIP2PVideo(u, scheduler="ddpm", num_ddim_steps=4)
    r1 = p1._runner(3, 8, 4, 4, 77)
    assert p1._runner(3, 8, 4, 4, 77) is r1 and p2._runner(3, 8, 4, 4, 77) is r1 and len(made) == 1  # shared across pipes
    assert p1._runner(3, 8, 4, 4, 77, slot=1) is not r1 and len(made) == 2                              # one per concurrency slot
    u.weights_version = 2
    assert p1._runner(3, 8, 4, 4, 77) is not r1 and len(made) == 3
    assert all(k[1] == 2 for k in inf._RUNNERS if k[0] == id(u))  # the stale graphs are gone
    uid = id(u)
    del p1, p2, u
    made.clear()
    gc.collect()
    assert not [k for k in inf._RUNNERS if k[0] == uid]            # and everything dies with the UNet
    # ... except when the collector runs inside a stream capture (no device synchronize allowed there): the id is parked
    # and the graphs are destroyed by the next shared_runner() call
    u = FakeUNet()
    inf.InferenceIP2PVideo(u, scheduler="ddim", num_ddim_steps=2)._runner(3, 8, 4, 4, 77)
    uid, stale = id(u), list(inf._RUNNERS.values())
    monkeypatch.setattr(inf, "_capturing", lambda: True)
    del u
    made.clear()
    gc.collect()
    assert uid in inf._DEAD and [k for k in inf._RUNNERS if k[0] == uid]
    monkeypatch.setattr(inf, "_capturing", lambda: False)
    u2 = FakeUNet()
    inf.InferenceIP2PVideo(u2, scheduler="ddim", num_ddim_steps=2)._runner(3, 8, 4, 4, 77)
    assert not inf._DEAD and not [r for r in inf._RUNNERS.values() if any(r is o for o in stale)]


def test_embed_tokens_rejects_bad_ids_with_hipkernelerror():
    from insv2v import ops, _lib
    src = open(ops.__file__).read()
    assert "raise HipKernelError" not in src and "_lib.HipKernelError(\"embed.ids" in src
    assert issubclass(_lib.HipKernelError, Exception)


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


def test_bench_clip_groups():
    """bench.py's throughput mode: the K timed steps are split into as few, as even groups as possible of <= 20 clips in flight
    (<= 10 with cap = 10, the round-3 schedule)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    want = {1: (1, [1]), 2: (1, [1, 1]), 3: (3, [3]), 4: (4, [4]), 5: (5, [5]), 6: (6, [6]), 10: (10, [10]), 11: (6, [6, 5]), 12: (6, [6, 6]),
            20: (10, [10, 10]), 25: (9, [9, 8, 8])}
    for k, exp in want.items():
        assert bench.clip_groups(k, 0, cap=10) == exp, (k, bench.clip_groups(k, 0, cap=10))
    want20 = {5: (5, [5]), 12: (12, [12]), 20: (20, [20]), 21: (11, [11, 10]), 25: (13, [13, 12]), 45: (15, [15, 15, 15])}
    for k, exp in want20.items():
        assert bench.clip_groups(k, 0) == exp, (k, bench.clip_groups(k, 0))
    for k in range(1, 40):
        for c in (0, 1, 2, 4, 7):
            cc, sizes = bench.clip_groups(k, c)
            assert sum(sizes) == k and max(sizes) <= cc and min(sizes) >= 1
    assert bench.clip_groups(8, 0, plain=False) == (1, [1] * 8)
    assert bench.clip_groups(5, 2) == (2, [2, 2, 1])
    assert bench.max_clips_in_flight() == 20 and bench.max_clips_in_flight(24, 48, 64) == 7 and bench.clip_groups(20, 0, cap=5) == (5, [5, 5, 5, 5])


def test_bench_top_shapes():
    """roofline.top_shapes: the dominant single launch shapes of the GEMM family, each with its own rate (bench.py)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lin = ("lin", 368640, 5120, 640, 1, 2, False)
    recs = ([("gemm_kernel", 2.0 * 368640 * 5120 * 640, 2.0e-3, lin)] * 10 + [("gemm_kernel", 3.6e12, 4.0e-3, ("ffn", 1474560, 320, 1280))] * 10
            + [("attn_kernel", 1e12, 9e-3, ("attn", 960, 8, 40, 1536, 1536))] * 4 + [("gemm_kernel", 1.0, 1.0, None)])
    top = bench.top_shapes(recs, n=2)
    assert [t["shape"][0] for t in top] == ["ffn", "lin"] and top[1]["shape"] == list(lin) and top[0]["launches"] == 10
    assert abs(top[0]["achieved"] - 900.0) < 1e-6 and abs(top[0]["frac"] - 0.36) < 1e-9 and abs(top[1]["avg_launch_us"] - 2000.0) < 1e-6
    assert abs(top[1]["algorithmic_gbytes_per_launch"] - bench.algorithmic_bytes(lin) / 1e9) < 1e-3
    json.dumps(top)


def test_bench_self_launch_command_and_no_gpu_exit():
    """`python bench.py --gpus N` without RANK in the environment starts its own ranks: the command it re-executes
    as is the driver's own N > 1 form; on a box without N GPUs it says so and exits (no assert about WORLD_SIZE, no CPU fallback)."""
    import importlib.util, subprocess
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    port = int(bench.self_launch_command(2, [])[bench.self_launch_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "AssertionError" not in r.stderr, r.stderr[-500:]


def test_xattn_fragment_streams_compute_the_cross_attention_block():
    """fused.pack_xattn_stream / pack_xattn_kv against a lane-level emulation of the schedule insv2v_xattn_fused runs (csrc/fused_rows.hip xa_op):
    v_mfma_f32_32x32x16_f16 operand / accumulator layouts, the C-layout -> operand chaining, head masking inside the K / V fragments."""
    import torch
    from insv2v import fused
    torch.manual_seed(0)
    C, H, L, D = 320, 8, 77, 40
    lane = torch.arange(64)
    col, half = lane & 31, lane >> 5

    def mfma(a, b, acc):            # a, b: [64, 8] fragments; acc [32 rows, 32 cols] += A . B^T
        A, B = torch.zeros(32, 16), torch.zeros(32, 16)
        for jj in range(8):
            A[col, 8 * half + jj] = a[:, jj]
            B[col, 8 * half + jj] = b[:, jj]
        return acc + A @ B.T

    def pack_tile(acc):             # C layout -> two operand fragments (registers 0..7 and 8..15 of every lane)
        out = []
        for u in range(2):
            f = torch.zeros(64, 8)
            for jj in range(8):
                r = 8 * u + jj
                f[:, jj] = acc[(r & 3) + 8 * (r >> 2) + 4 * half, col]
            out.append(f.half().float())
        return out

    x = torch.randn(32, C)
    xn = torch.nn.functional.layer_norm(x, (C,)).half().float()
    wq, wo = (torch.randn(C, C) * C ** -0.5).half().float(), (torch.randn(C, C) * C ** -0.5).half().float()
    bq, bo = torch.randn(C) * 0.1, torch.randn(C) * 0.1
    kv = torch.randn(L, 2 * C).half()
    w = fused.pack_xattn_stream(wq, bq, wo, bo).float().reshape(-1, 64, 8)
    ks = fused.pack_xattn_kv(kv, 1, L, C, H)[0].float().reshape(-1, 64, 8)
    assert w.shape[0] == fused.XA_Q_FR + fused.XA_O_FR and ks.shape[0] == fused.XA_KV_FR
    stream = torch.cat([w[:fused.XA_Q_FR], ks, w[fused.XA_Q_FR:]], 0)

    ones = torch.zeros(64, 8)
    ones[:32, 0] = ones[:32, 1] = 1.0
    xf = [torch.stack([xn[col, 16 * s + 8 * half + e] for e in range(8)], 1) for s in range(20)]
    qs, afr = [None] * 20, [None] * 20
    scale = D ** -0.5
    acc = {}
    out = torch.zeros(32, C)
    f = 0
    for p in range(5):              # Q section: tile pairs interleaved over 21 k-steps
        a0, a1 = torch.zeros(32, 32), torch.zeros(32, 32)
        for s in range(21):
            b = xf[s] if s < 20 else ones
            a0 = mfma(stream[f], b, a0); a1 = mfma(stream[f + 1], b, a1); f += 2
        qs[4 * p], qs[4 * p + 1] = pack_tile(a0)
        qs[4 * p + 2], qs[4 * p + 3] = pack_tile(a1)
    f = fused.XA_Q_FR
    for gh in range(8):
        G, h = gh // 4, gh % 4
        S = [torch.zeros(32, 32) for _ in range(3)]
        for st in range(3):
            for kt in range(3):
                S[kt] = mfma(stream[f], qs[10 * G + fused._xa_kstep(h, st)], S[kt]); f += 1
        s_all = torch.cat(S, 0)                                 # [96 keys, 32 tokens]
        s_all[L:] = -1e30
        pr = torch.softmax(s_all * scale, dim=0)
        P = [t for kt in range(3) for t in pack_tile(pr[32 * kt:32 * kt + 32])]
        if h == 0:
            O = [torch.zeros(32, 32) for _ in range(5)]
        for kst in range(6):
            for sel in range(2):
                t = (40 * h) // 32 + sel
                O[t] = mfma(stream[f], P[kst], O[t]); f += 1
        if h == 3:
            for t in range(5):
                afr[2 * (5 * G + t)], afr[2 * (5 * G + t) + 1] = pack_tile(O[t])
    f = fused.XA_Q_FR + fused.XA_KV_FR
    for p in range(5):
        a0, a1 = torch.zeros(32, 32), torch.zeros(32, 32)
        for s in range(21):
            b = afr[s] if s < 20 else ones
            a0 = mfma(stream[f], b, a0); a1 = mfma(stream[f + 1], b, a1); f += 2
        out[:, 64 * p:64 * p + 32], out[:, 64 * p + 32:64 * p + 64] = a0.T, a1.T
    out = out + x
    # reference: diffusers Attention on LayerNorm(x) with the text K / V
    q = (xn @ wq.T + bq).reshape(32, H, D)
    k, v = kv[:, :C].float().reshape(L, H, D), kv[:, C:].float().reshape(L, H, D)
    att = torch.softmax(torch.einsum("thd,lhd->htl", q, k) * scale, -1)
    ref = x + torch.einsum("htl,lhd->thd", att, v).reshape(32, C) @ wo.T + bo
    err = (out - ref).abs().max().item()
    assert err < 1e-3 * ref.abs().max().item(), err   # measured 9e-5: fp16 roundings of q, P and the attention output only


def test_xattn640_fragment_streams_compute_the_cross_attention():
    """fused.pack_xattn_q_stream / pack_xattn640_kv against a lane-level emulation of the schedule insv2v_xattn_attn runs (csrc/fused_rows.hip
    xb_op): per 160-channel head group the q tiles, then per head 15 K and 18 V fragments; heads of 80 channels = 5 whole k-steps."""
    import torch
    from insv2v import fused
    torch.manual_seed(1)
    C, H, L, D = 640, 8, 77, 80
    lane = torch.arange(64)
    col, half = lane & 31, lane >> 5

    def mfma(a, b, acc):
        A, B = torch.zeros(32, 16), torch.zeros(32, 16)
        for jj in range(8):
            A[col, 8 * half + jj] = a[:, jj]
            B[col, 8 * half + jj] = b[:, jj]
        return acc + A @ B.T

    def pack_tile(acc):
        out = []
        for u in range(2):
            f = torch.zeros(64, 8)
            for jj in range(8):
                r = 8 * u + jj
                f[:, jj] = acc[(r & 3) + 8 * (r >> 2) + 4 * half, col]
            out.append(f.half().float())
        return out

    x = torch.randn(32, C)
    xn = torch.nn.functional.layer_norm(x, (C,)).half().float()
    wq, bq = (torch.randn(C, C) * C ** -0.5).half().float(), torch.randn(C) * 0.1
    kv = torch.randn(L, 2 * C).half()
    w = fused.pack_xattn_q_stream(wq, bq).float().reshape(4, fused.XB_Q_FR, 64, 8)
    ks = fused.pack_xattn640_kv(kv, 1, L, C, H)[0].float().reshape(4, fused.XB_KV_FR, 64, 8)
    ones = torch.zeros(64, 8)
    ones[:32, 0] = ones[:32, 1] = 1.0
    xf = [torch.stack([xn[col, 16 * s + 8 * half + e] for e in range(8)], 1) for s in range(40)]
    scale = D ** -0.5
    out = torch.zeros(32, C)
    for G in range(4):
        qs, f = [None] * 10, 0
        for p in range(2):
            a0, a1 = torch.zeros(32, 32), torch.zeros(32, 32)
            for s in range(41):
                b = xf[s] if s < 40 else ones
                a0 = mfma(w[G, f], b, a0); a1 = mfma(w[G, f + 1], b, a1); f += 2
            qs[4 * p], qs[4 * p + 1] = pack_tile(a0)
            qs[4 * p + 2], qs[4 * p + 3] = pack_tile(a1)
        a0 = torch.zeros(32, 32)
        for s in range(41):
            a0 = mfma(w[G, f], xf[s] if s < 40 else ones, a0); f += 1
        qs[8], qs[9] = pack_tile(a0)
        O, f = [torch.zeros(32, 32) for _ in range(5)], 0
        for h in range(2):
            S = [torch.zeros(32, 32) for _ in range(3)]
            for st in range(5):
                for kt in range(3):
                    S[kt] = mfma(ks[G, f], qs[5 * h + st], S[kt]); f += 1
            s_all = torch.cat(S, 0)
            s_all[L:] = -1e30
            pr = torch.softmax(s_all * scale, dim=0)
            P = [t for kt in range(3) for t in pack_tile(pr[32 * kt:32 * kt + 32])]
            for kst in range(6):
                for sel in range(3):
                    O[2 * h + sel] = mfma(ks[G, f], P[kst], O[2 * h + sel]); f += 1
        for t in range(5):
            out[:, 160 * G + 32 * t:160 * G + 32 * t + 32] = O[t].T
    q = (xn @ wq.T + bq).reshape(32, H, D)
    k, v = kv[:, :C].float().reshape(L, H, D), kv[:, C:].float().reshape(L, H, D)
    att = torch.softmax(torch.einsum("thd,lhd->htl", q, k) * scale, -1)
    ref = torch.einsum("htl,lhd->thd", att, v).reshape(32, C)
    err = (out - ref).abs().max().item()
    assert err < 2e-3 * ref.abs().max().item(), err


def test_tattn640_fragment_stream_computes_the_temporal_attention():
    """fused.pack_tattn_qkv_stream against a lane-level emulation of the schedule insv2v_tattn_attn runs (csrc/fused_rows.hip tb_op): a wave = 2
    pixels x 16 frames; per 160-channel group the q / k tiles (weights as the A operand), S^T = K . Q^T with the two pixels on the diagonal
    16x16 blocks, V with the operands swapped so its packed tile is the A operand of O^T = V^T . P^T; per-frame bias through the one-hot k-step."""
    import torch
    from insv2v import fused
    torch.manual_seed(2)
    C, H, F_, D = 640, 8, 16, 80
    lane = torch.arange(64)
    col, half = lane & 31, lane >> 5

    def mfma(a, b, acc):            # acc [rows of a, rows of b] += A . B^T
        A, B = torch.zeros(32, 16), torch.zeros(32, 16)
        for jj in range(8):
            A[col, 8 * half + jj] = a[:, jj]
            B[col, 8 * half + jj] = b[:, jj]
        return acc + A @ B.T

    def pack_tile(acc):
        out = []
        for u in range(2):
            f = torch.zeros(64, 8)
            for jj in range(8):
                r = 8 * u + jj
                f[:, jj] = acc[(r & 3) + 8 * (r >> 2) + 4 * half, col]
            out.append(f.half().float())
        return out

    x = torch.randn(2, F_, C)                                   # [pixel, frame, C]; wave token = 16 * pixel + frame
    xn = torch.nn.functional.layer_norm(x, (C,)).half().float().reshape(32, C)
    wqkv = (torch.randn(3 * C, C) * C ** -0.5).half().float()
    table = (torch.randn(F_, 3 * C) * 0.3).half().float()
    st = fused.pack_tattn_qkv_stream(wqkv, table).float().reshape(4, 624, 64, 8)
    xf = [torch.stack([xn[col, 16 * s + 8 * half + e] for e in range(8)], 1) for s in range(40)]
    fr, pp = col & 15, col >> 4
    fhot = torch.zeros(64, 8)
    for e in range(8):
        fhot[:, e] = ((half == (fr >> 3)) & (e == (fr & 7))).float()
    scale = D ** -0.5
    out = torch.zeros(32, C)
    for G in range(4):
        qs, ks, f = [None] * 10, [None] * 10, 0
        for tl in range(5):
            aq, ak = torch.zeros(32, 32), torch.zeros(32, 32)
            for s in range(41):
                b = xf[s] if s < 40 else fhot
                aq = mfma(st[G, f], b, aq); ak = mfma(st[G, f + 1], b, ak); f += 2
            qs[2 * tl], qs[2 * tl + 1] = pack_tile(aq)
            ks[2 * tl], ks[2 * tl + 1] = pack_tile(ak)
        PB, invl = [], []
        for h in range(2):
            S = torch.zeros(32, 32)                               # [key token, query token]
            for s5 in range(5):
                S = mfma(ks[5 * h + s5], qs[5 * h + s5], S)
            same = (torch.arange(32)[:, None] >> 4) == (torch.arange(32)[None, :] >> 4)
            e = torch.exp((S - torch.where(same, S, torch.tensor(-1e30)).max(0).values[None, :]) * scale) * same
            e = e.half().float()
            invl.append(1.0 / e.sum(0))
            PB.append(pack_tile(e))                               # B operand: rows = query tokens, k = key tokens in C-layout order
        tiles = []
        for grp in ((0, 1), (2, 3), (4,)):
            accs = [torch.zeros(32, 32) for _ in grp]
            for s in range(41):
                a = xf[s] if s < 40 else fhot
                for i in range(len(grp)):
                    accs[i] = mfma(a, st[G, f], accs[i]); f += 1  # operands swapped: [token, channel]
            tiles += accs
        assert f == 615
        for tl, accV in enumerate(tiles):
            v0, v1 = pack_tile(accV)                              # A operand of O^T: rows = channels, k = tokens
            o = torch.zeros(32, 32)
            for qd in range(4):
                h = (32 * tl + 8 * qd) // 80
                O = mfma(v1, PB[h][1], mfma(v0, PB[h][0], torch.zeros(32, 32)))
                rows = [(r & 3) + 8 * (r >> 2) + 4 * hh for hh in range(2) for r in range(4 * qd, 4 * qd + 4)]
                o[rows] = O[rows] * invl[h][None, :]
            out[:, 160 * G + 32 * tl:160 * G + 32 * tl + 32] = o.T
    frame = torch.arange(32) & 15
    qkv = (xn @ wqkv.T + table[frame]).half().float().reshape(2, F_, 3, H, D)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))        # [pixel, head, frame, d]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(32, C)
    err = (out - ref).abs().max().item()
    assert err < 3e-3 * ref.abs().max().item(), err


def test_single_video_dataset_sampling_plan_and_items(tmp_path):
    """SingleVideoDataset (dataset/single_video_dataset.py:10-117): the frame-sampling plan against values worked out by hand
    from the reference's formulas, and items read from a frame directory: clip length, frame stride, aspect-preserving resize + centre
    crop, value range, last-frame repetition at the end of the video."""
    import numpy as np
    from PIL import Image
    from insv2v.video_io import SingleVideoDataset, sampling_plan
    # video 30 fps, 100 frames; sampling_fps 24 -> min(24, 30) = 24, gap = int(30 / 24) = 1; 16 frames per item; 100 - 1 * 15 = 85 starts
    assert sampling_plan(30.0, 100, 24, 0, 16) == (24, 1, 16, 85)
    # sampling_fps 8 -> gap int(30 / 8) = 3; items of min(16, 100 // 3) = 16 frames; 100 - 3 * 15 = 55 starts
    assert sampling_plan(29.97, 100, 8, 0, 16) == (8, 3, 16, 55)
    # sampling_fps None: frame_gap decides: fps 30 // (1 + 2) = 10, min(16, 20 // 2) = 10 frames, 20 - 2 * 9 = 2 starts
    assert sampling_plan(30.0, 20, None, 2, 16) == (10, 2, 10, 2)
    d = tmp_path / "clip"
    d.mkdir()
    for i in range(12):
        arr = np.full((40, 80, 3), i * 20, dtype=np.uint8)      # 2 : 1 frames, constant colour = frame index
        arr[:, :8] = 255                                         # a white stripe on the left edge: cropped away by the centre crop
        Image.fromarray(arr).save(d / f"{i:03d}.png")
    (d / "fps.txt").write_text("12")
    ds = SingleVideoDataset(str(d), "a clip", sampling_fps=4, num_frames=4, output_size=(32, 32))
    assert (ds.sampling_fps, ds.frame_gap, ds.num_frames, len(ds)) == (4, 3, 4, 3) and ds.video_id == "clip"
    item = ds[1]
    assert item["frames"].shape == (4, 3, 32, 32) and item["text"] == "a clip" and int(item["fps"]) == 4
    want = [(1 + 3 * i) * 20 / 127.5 - 1.0 for i in range(4)]   # frames 1, 4, 7, 10
    got = item["frames"][:, 0, 16, 16].tolist()
    assert all(abs(g - w) < 1e-2 for g, w in zip(got, want)), (got, want)
    assert float(item["frames"].max()) <= 1.0 and float(item["frames"][..., 0].max()) < 0.99   # the stripe (x < 8 of 80 -> < 6.4 of 64) is outside the crop
    last = ds[2]["frames"]                                      # frames 2, 5, 8, 11: all present
    assert abs(float(last[3, 0, 0, 0]) - (11 * 20 / 127.5 - 1.0)) < 1e-2


def test_fit_frame_portrait_is_padded_on_both_sides():
    """a narrow (portrait) frame (single_video_dataset.py:90-93).  torchvision's ``F.pad(frame, (margin, 0))`` reads a length-2
    padding as (left/right, top/bottom): ``margin`` columns of the value 0 (before normalisation, i.e. -1 after it) on BOTH sides, the
    picture centred, the item target_w + 2 * margin wide (= W when W - target_w is even, one column short otherwise, as in the reference)."""
    from insv2v.video_io import fit_frame
    frame = torch.full((3, 80, 40), 200, dtype=torch.uint8)          # 1 : 2 portrait
    out = fit_frame(frame, (32, 32))                                  # target_w = int(32 * 0.5) = 16, margin = 8
    assert out.shape == (3, 32, 32)
    assert torch.all(out[:, :, :8] == -1.0) and torch.all(out[:, :, 24:] == -1.0)
    assert torch.allclose(out[:, :, 8:24], torch.full((3, 32, 16), 200 / 127.5 - 1.0), atol=1e-6)
    out = fit_frame(torch.full((3, 80, 42), 100, dtype=torch.uint8), (32, 32))   # target_w = int(16.8) = 16: still margin 8
    assert out.shape == (3, 32, 32)
    out = fit_frame(torch.full((3, 64, 30), 100, dtype=torch.uint8), (32, 32))   # target_w = 15, margin = 8: 15 + 16 = 31 columns
    assert out.shape == (3, 32, 31) and torch.all(out[:, :, :8] == -1.0) and torch.all(out[:, :, 23:] == -1.0)


def test_edit_videos_stacks_optical_flow_units():
    """optical-flow units no longer fall back to one clip at a time - window k of ALL units is ONE run_stacked call whose
    entries carry the unit's flow source: precomputed ``flows`` (flows_per_window[k]) or the last R frames of the previous window + the new
    frames for the pipe's estimator (insv2v_run_loveu_tgve.py:141-147)."""
    from insv2v.run_loveu_tgve import edit_videos

    class FakeModel:
        scale_factor = 0.5

        class unet:
            device = "cpu"

        def encode_image_to_latent(self, frames, noise=None):
            return torch.zeros(1, frames.shape[1], 4, 2, 2)

        def decode_latent_to_image(self, lat):
            return torch.zeros(1, lat.shape[1], 3, 16, 16)

    class FlowPipe:
        flow_estimator = staticmethod(lambda q, r: torch.zeros(len(r), 2, 16, 16))

        def __init__(self):
            self.stacks = []

        def obtain_flow_batched(self, *a):
            raise AssertionError("the stack resolves the flows, not the driver")

        def run_stacked(self, calls):
            self.stacks.append(calls)
            return [{"latent": c["latent"]} for c in calls]

    frames = [torch.full((1, 32, 3, 16, 16), float(u)) for u in range(3)]
    frames[1][:, 12:16] = 7.0     # the last R = 4 frames of unit 1's first window
    pipe = FlowPipe()
    flows1 = [[torch.zeros(4, 2, 16, 16)] * 12, [torch.zeros(12, 2, 16, 16)] * 4]
    units = [dict(frames=frames[0], text_cond=None, text_uncond=None),
             dict(frames=frames[1], text_cond=None, text_uncond=None),
             dict(frames=frames[2], text_cond=None, text_uncond=None, flows_per_window=flows1)]
    outs = edit_videos(FakeModel(), pipe, units)
    assert len(outs) == 3 and all(o.shape == (1, 32, 3, 16, 16) for o in outs)
    assert [len(c) for c in pipe.stacks] == [3, 3, 3]                       # windows 16 / 12 / 4 new frames, all units per stack
    first, second, third = pipe.stacks
    assert all("latent_ref" not in c and "flows" not in c and "ref_images" not in c for c in first)
    assert [tuple(c["latent_ref"].shape[1:2]) for c in second] == [(4,)] * 3 and [tuple(c["latent_ref"].shape[1:2]) for c in third] == [(12,)] * 3
    # units 0, 1: the estimator's inputs; unit 2: its precomputed flows, no images
    assert tuple(second[0]["ref_images"].shape) == (1, 4, 3, 16, 16) and tuple(second[0]["query_images"].shape) == (1, 12, 3, 16, 16)
    assert float(second[1]["ref_images"].min()) == 7.0 and float(second[1]["query_images"].max()) == 1.0
    assert tuple(third[1]["ref_images"].shape) == (1, 12, 3, 16, 16) and tuple(third[1]["query_images"].shape) == (1, 4, 3, 16, 16)
    assert second[2]["flows"] is flows1[0] and third[2]["flows"] is flows1[1] and "ref_images" not in second[2]
    nof = FlowPipe()
    nof.flow_estimator = None
    with pytest.raises(RuntimeError, match="flow source"):
        edit_videos(FakeModel(), nof, units[:2])
