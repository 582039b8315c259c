"""Host-side video I/O of the LOVEU-TGVE driver: the dataset reader and the GIF / JPG writers the
reference driver calls around the hot path.  Pure host code (no kernels): strings, files, PIL.

  LoveuTgveVideoDataset   dataset/loveu_tgve_dataset.py:10-100  (same CSV grammar, same item dict)
  SingleVideoDataset      dataset/single_video_dataset.py:10-117 (frame sampling plan, aspect-preserving resize + centre crop / pad, item dict)
  save_tensor_to_gif      misc_utils/image_utils.py:127-132,233-235
  save_tensor_to_images   misc_utils/image_utils.py:237-241
  output_paths            insv2v_run_loveu_tgve.py:104-114 (result folder / file naming)

Decoding: the reference reads ``<video>.mp4`` with OpenCV.  cv2 is not installed in this image, so it is used when
importable and otherwise a directory ``<source_folder>/<video_name>/`` of frame images (sorted *.jpg / *.png) is read with
PIL; PIL's bilinear resize stands in for ``cv2.resize`` (INTER_LINEAR) - pixel values can differ in the last bit, which is
an input difference, not a difference of the editing path.
"""
import csv
import os

import numpy as np
import torch

_IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp")


def _to_tensor(rgb_uint8):
    """transforms.ToTensor + Normalize(0.5, 0.5): HWC uint8 -> CHW float in [-1, 1] (loveu_tgve_dataset.py:14-17)."""
    t = torch.from_numpy(np.array(rgb_uint8, dtype=np.uint8)).permute(2, 0, 1).float() / 255.0
    return (t - 0.5) / 0.5


class LoveuTgveVideoDataset:
    def __init__(self, root_dir, image_size=(480, 480)):
        self.root_dir, self.image_size = root_dir, tuple(image_size)
        self.data = {}
        source_folder = None
        with open(os.path.join(root_dir, "LOVEU-TGVE-2023_Dataset.csv"), "r") as f:
            reader = csv.reader(f)
            next(reader, None)  # header
            for row in reader:
                if not row or len(row[0]) == 0:
                    continue
                if row[0].endswith("Videos:"):  # section header, e.g. "DAVIS Videos:" / "Youtube Videos:"
                    kind = row[0].split(" ")[0]
                    source_folder = (kind if kind == "DAVIS" else kind.lower()) + "_480p/480p_videos"
                elif len(row) > 1:
                    self.data[row[0]] = dict(video_name=row[0], original=row[1], style=row[2], object=row[3], background=row[4],
                                             multiple=row[5], source_folder=source_folder)

    def _paths(self, video_name, source_folder):
        base = os.path.join(self.root_dir, source_folder, video_name)
        return base + ".mp4", base

    def load_frames(self, video_name, source_folder):
        mp4, frame_dir = self._paths(video_name, source_folder)
        frames = []
        try:
            import cv2
        except ImportError:
            cv2 = None
        if cv2 is not None and os.path.exists(mp4):
            cap = cv2.VideoCapture(mp4)
            while cap.isOpened():
                ret, frame = cap.read()
                if not ret:
                    break
                frame = cv2.cvtColor(cv2.resize(frame, self.image_size), cv2.COLOR_BGR2RGB)
                frames.append(_to_tensor(frame))
            cap.release()
        elif os.path.isdir(frame_dir):
            from PIL import Image
            for name in sorted(n for n in os.listdir(frame_dir) if n.lower().endswith(_IMG_EXT)):
                img = Image.open(os.path.join(frame_dir, name)).convert("RGB").resize(self.image_size, Image.BILINEAR)
                frames.append(_to_tensor(np.asarray(img)))
        else:
            raise FileNotFoundError(f"{mp4} needs OpenCV (not installed) and no frame directory {frame_dir} exists")
        if not frames:
            raise RuntimeError(f"no frames decoded for {video_name}")
        return torch.stack(frames, dim=0)

    def load_fps(self, video_name, source_folder):
        mp4, frame_dir = self._paths(video_name, source_folder)
        try:
            import cv2
            if os.path.exists(mp4):
                cap = cv2.VideoCapture(mp4)
                fps = cap.get(cv2.CAP_PROP_FPS)
                cap.release()
                return fps
        except ImportError:
            pass
        fps_file = os.path.join(frame_dir, "fps.txt")
        return float(open(fps_file).read()) if os.path.exists(fps_file) else 0.0

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        video_name = idx if isinstance(idx, str) else list(self.data.keys())[idx]
        item = self.data[video_name].copy()
        item["frames"] = self.load_frames(video_name, item["source_folder"])
        item["fps"] = self.load_fps(video_name, item["source_folder"])
        return item


def sampling_plan(video_fps, total_frames, sampling_fps=24, frame_gap=0, num_frames=2):
    """The frame-sampling arithmetic of SingleVideoDataset.__init__ (single_video_dataset.py:36-58), host integers only:
    returns (sampling_fps, frame_gap, frames per item, number of valid start frames).  ``sampling_fps`` may be an int, a list (the reference
    draws one at random: pass the drawn value for a reproducible plan) or None (then ``frame_gap`` decides)."""
    video_fps = round(video_fps)
    if sampling_fps is not None:
        if isinstance(sampling_fps, (list, tuple)):
            import random
            sampling_fps = random.choice(list(sampling_fps))
        if not isinstance(sampling_fps, int):
            raise ValueError(f"sampling_fps should be int or list of int, got {sampling_fps}")
        sampling_fps = int(min(sampling_fps, video_fps))
        frame_gap = max(0, int(video_fps / sampling_fps))
    else:
        sampling_fps = video_fps // (1 + frame_gap)
    n = min(num_frames, total_frames // frame_gap)      # (a zero frame_gap divides by zero in the reference too)
    starts = max(0, total_frames - frame_gap * (n - 1))
    return sampling_fps, frame_gap, n, starts


def fit_frame(frame_chw_uint8, output_size):
    """One decoded RGB frame [3, h, w] uint8 -> [3, H, W] float in [-1, 1] (single_video_dataset.py:82-96): resize to height H keeping the
    aspect ratio (bilinear, antialiased, like torchvision's tensor resize), then centre-crop the width to W, or - narrow (portrait) frames -
    pad ``margin`` zero-valued (pre-normalisation) columns on BOTH sides: torchvision's ``F.pad(frame, (margin, 0))`` reads a length-2
    padding as (left/right, top/bottom), so the item is target_w + 2 * margin wide with the picture centred."""
    W, H = output_size[0], output_size[1]
    _, h, w = frame_chw_uint8.shape
    target_w = int(W * (w / h))
    x = torch.nn.functional.interpolate(frame_chw_uint8[None].float(), size=(H, target_w), mode="bilinear", antialias=True, align_corners=False)[0]
    x = x.round().clamp(0, 255)                           # torchvision rounds back to uint8
    if target_w > H:
        margin = (target_w - H) // 2
        x = x[:, :H, margin:margin + W]
    else:
        margin = (H - target_w) // 2
        x = torch.nn.functional.pad(x, (margin, margin, 0, 0))  # torchvision F.pad(frame, (margin, 0)): left AND right by margin
    return x / 127.5 - 1.0


class SingleVideoDataset:
    """One video as a dataset of ``num_frames``-frame clips starting at every valid frame (dataset/single_video_dataset.py:10-117; the
    gradio demo's and the trainer's loader).  ``video_file`` is an .mp4 (needs OpenCV) or a DIRECTORY of frame images with an optional
    ``fps.txt`` (this image has no OpenCV); items are ``dict(frames [n,3,H,W] in [-1,1], video_id, text, fps)`` like the reference's."""

    def __init__(self, video_file, video_description, sampling_fps=24, frame_gap=0, num_frames=2, output_size=(512, 512), mode="train"):
        self.video_file, self.description, self.output_size, self.mode = video_file, video_description, tuple(output_size), mode
        self.video_id = os.path.splitext(os.path.basename(os.path.normpath(video_file)))[0]
        self._names = None
        if os.path.isdir(video_file):
            self._names = sorted(n for n in os.listdir(video_file) if n.lower().endswith(_IMG_EXT))
            fps_file = os.path.join(video_file, "fps.txt")
            video_fps, total = (float(open(fps_file).read()) if os.path.exists(fps_file) else 24.0), len(self._names)
        else:
            import cv2   # as the reference: no fallback for a container format without OpenCV
            cap = cv2.VideoCapture(video_file)
            video_fps, total = cap.get(cv2.CAP_PROP_FPS), int(cap.get(cv2.CAP_PROP_FRAME_COUNT))
            cap.release()
        self.total_frames = total
        self.sampling_fps, self.frame_gap, self.num_frames, self.total_possible_starting_frames = sampling_plan(
            video_fps, total, sampling_fps, frame_gap, num_frames)

    def __len__(self):
        return self.total_possible_starting_frames

    def _read(self, idx):
        if self._names is not None:
            if idx >= len(self._names):
                return None
            from PIL import Image
            return torch.from_numpy(np.asarray(Image.open(os.path.join(self.video_file, self._names[idx])).convert("RGB")).copy()).permute(2, 0, 1)
        import cv2
        cap = cv2.VideoCapture(self.video_file)
        cap.set(cv2.CAP_PROP_POS_FRAMES, idx)
        ret, frame = cap.read()
        cap.release()
        return None if frame is None else torch.from_numpy(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)).permute(2, 0, 1)

    def __getitem__(self, index):
        frames = []
        if self.total_frames > 1 + self.frame_gap:
            for i in range(self.num_frames):
                f = self._read(index + i * self.frame_gap)
                if f is not None:
                    frames.append(fit_frame(f, self.output_size))
        while len(frames) < self.num_frames:   # short reads repeat the last frame (:104-106)
            frames.append(frames[-1])
        return {"frames": torch.stack(frames[:self.num_frames], dim=0), "video_id": self.video_id, "text": self.description,
                "fps": torch.tensor(self.sampling_fps)}


def _to_uint8_frames(images):
    """[1, T, 3, H, W] in [-1, 1] -> list of HWC uint8 (image_utils.py:234 then :130: truncating cast)."""
    arr = images.squeeze(0).detach().float().cpu().numpy().transpose(0, 2, 3, 1) / 2 + 0.5
    return [(np.clip(a, 0.0, 1.0) * 255).astype(np.uint8) for a in arr]


def save_tensor_to_gif(images, filename, fps):
    from PIL import Image
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    frames = [Image.fromarray(a) for a in _to_uint8_frames(images)]
    frames[0].save(filename, save_all=True, append_images=frames[1:], duration=int(round(1000.0 / fps)), loop=0)


def save_tensor_to_images(images, output_dir):
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    for i, a in enumerate(_to_uint8_frames(images)):
        Image.fromarray(a).save(f"{output_dir}/{i:03d}.jpg")


def output_paths(prompt_source, image_size, video_id, video_cfg, text_cfg, num_frames, video_name, prompt_key, final_prompt):
    """(gif path, image dir) exactly as insv2v_run_loveu_tgve.py:104-114 lays the results out."""
    tag = {"edit": "edit_prompt", "original": "original_prompt"}[prompt_source]
    cfg = f"VIDEO_CFG_{video_cfg}_TEXT_CFG_{text_cfg}"
    out_folder = f"v2v_results/{tag}/loveu_tgve_{image_size}/gif/VID_{video_id}/{cfg}"
    image_dir = f"v2v_results/{tag}/loveu_tgve_{image_size}/images_{num_frames}/{cfg}/{video_name}/{prompt_key}"
    text = "_".join(final_prompt.split(" "))
    return f"{out_folder}/{prompt_key}_{num_frames}_{text}.gif", image_dir
