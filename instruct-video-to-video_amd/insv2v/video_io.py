"""Host-side video I/O of the LOVEU-TGVE driver: the dataset reader and the GIF / JPG writers the
reference driver calls around the hot path.  Pure host code (no kernels): strings, files, PIL.

  LoveuTgveVideoDataset   dataset/loveu_tgve_dataset.py:10-100  (same CSV grammar, same item dict)
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
