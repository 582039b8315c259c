#!/usr/bin/env python3
"""Pin oracle/raft.py against the real torchvision ``raft_large``.

torchvision is NOT installed in the build image, so the optical-flow oracle is this build's reading of the published network ("parity
unpinned" for row f3).  On any machine that has torchvision (>= 0.13) this ONE command pins it:

    python tools/pin_raft_oracle.py            # -> tests/golden/raft_large_pin.npz + a max-difference report

It builds ``torchvision.models.optical_flow.raft_large(weights=None)``, loads the SAME key-hashed state dict the tests use
(insv2v.synth.synth_raft_state_dict: torchvision's exact key set, BatchNorm running statistics that are not the identity), runs the real
network and the oracle on the same seeded frame pairs, asserts agreement, and writes the real network's flows as a fixture that
tests/test_cpu_oracle.py::test_raft_oracle_vs_torchvision_pin (skipped while the file is absent) and the GPU test check from then on.
With ``--weights`` it also runs the published checkpoint (``Raft_Large_Weights.DEFAULT`` state dict saved as a .pth) through both."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instruct-video-to-video_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=None, help="optional: a torchvision raft_large state dict (.pth) to run beside the key-hashed one")
    ap.add_argument("--size", type=int, nargs=2, default=[128, 192])
    a = ap.parse_args()
    try:
        from torchvision.models.optical_flow import raft_large
    except Exception as e:   # the build image: say what to do instead of failing obscurely
        print(f"torchvision is not importable here ({type(e).__name__}: {e}).\nRun this script where torchvision >= 0.13 is installed; "
              "it needs nothing else beyond this repository (CPU is enough).", file=sys.stderr)
        return 3
    import numpy as np
    import torch
    from insv2v import shapes, synth
    from oracle.raft import RAFTFlow as OracleFlow
    H, W = a.size
    out = {}
    cases = [("synth", synth.synth_raft_state_dict(shapes.raft_shapes()))]
    if a.weights:
        cases.append(("published", torch.load(a.weights, map_location="cpu")))
    for name, sd in cases:
        ref = raft_large(weights=None).eval()
        missing, unexpected = ref.load_state_dict(sd, strict=True), None
        ora = OracleFlow()
        ora.model.load_state_dict(sd)
        img1 = synth.synth_input("pin.img1", (2, 3, H, W), kind="uniform")
        img2 = synth.synth_input("pin.img2", (2, 3, H, W), kind="uniform")
        with torch.no_grad():
            # flow_utils.py:176-180: the preset maps [0,1] -> [-1,1]; frames are handed over as they are
            want = ref((img1 - 0.5) / 0.5, (img2 - 0.5) / 0.5, num_flow_updates=12)[-1]
            got = ora(img1, img2)
        d = (got - want).abs().max().item()
        print(f"[{name}] max |oracle - torchvision| = {d:.3e} px (|flow| max {want.abs().max().item():.2f})")
        assert d <= 1e-3 * max(1.0, want.abs().max().item()), "oracle/raft.py disagrees with torchvision raft_large"
        out[f"{name}_img1"], out[f"{name}_img2"], out[f"{name}_flow"] = img1.numpy(), img2.numpy(), want.numpy()
    path = os.path.join(ROOT, "tests", "golden", "raft_large_pin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
