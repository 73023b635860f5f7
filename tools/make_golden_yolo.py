"""Generate tests/golden/yolov8n.npz from the REFERENCE run on the one real exported graph it ships with weights
(examples/YOLOv8n_wasm/yolov8n_fp32, copied beside the oracle build by `make -C oracle ref`): seeded input, every 8th anchor column of the
[1,84,8400] output in fp16 and fp32 arithmetic plus whole-tensor sums (the fixture stays < 1 MB; the GPU test compares the FULL output
against the oracle run on the spot where oracle/_ref travelled, and against this subsample otherwise)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref as oref  # noqa: E402

YOLO = os.path.join(REPO, "oracle", "_ref", "yolov8n_fp32") + "/"


def yolo_input(seed=20):
    return np.random.default_rng(seed).random((1, 3, 640, 640), dtype=np.float32)     # pixels in [0,1), what the JS demo feeds


if __name__ == "__main__":
    assert oref.available() and os.path.exists(YOLO + "model.txt"), "make -C oracle ref"
    x = yolo_input()
    o16 = oref.run_model(YOLO, {"images": x}, fp16=True, threads=1)["output0"]
    o32 = oref.run_model(YOLO, {"images": x}, fp16=False, threads=1)["output0"]
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "yolov8n.npz"), seed=np.asarray(20), stride=np.asarray(8),
                        ref16=o16[:, :, ::8], ref32=o32[:, :, ::8], sum16=np.asarray(o16.astype(np.float64).sum()), sum32=np.asarray(o32.astype(np.float64).sum()),
                        abs16=np.asarray(np.abs(o16).astype(np.float64).sum()), abs32=np.asarray(np.abs(o32).astype(np.float64).sum()))
    print("yolov8n", o16.shape, "box drift", np.abs(o16[:, :4] - o32[:, :4]).max() / np.abs(o32[:, :4]).max(), "score drift", np.abs(o16[:, 4:] - o32[:, 4:]).max())
