"""Synthetic stand-in for an AudioSet-style data directory (there is no dataset in this image): N PCM16 clips of
`seconds` at 16 kHz (class-dependent tones + noise), train/val json lists and the label csv in the formats
/root/reference/src/dataloader.py reads.  Used to exercise / time `python -m aum.train` end to end."""
import argparse
import json
import os
import wave

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--val-clips", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--classes", type=int, default=527)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    rng = np.random.default_rng(0)
    mids = [f"/m/{i:05d}" for i in range(a.classes)]
    with open(os.path.join(a.out, "class_labels_indices.csv"), "w") as f:
        f.write("index,mid,display_name\n" + "".join(f'{i},{m},"class {i}"\n' for i, m in enumerate(mids)))
    n = int(a.seconds * 16000)
    t = np.arange(n) / 16000.0
    items = []
    for i in range(a.clips + a.val_clips):
        labs = rng.choice(a.classes, size=int(rng.integers(1, 4)), replace=False)
        x = 0.02 * rng.standard_normal(n)
        for c in labs:
            x += 0.2 * np.sin(2 * np.pi * (100.0 + 7000.0 * c / a.classes) * t + rng.random() * 6.28)
        p = os.path.join(a.out, f"clip{i:05d}.wav")
        with wave.open(p, "wb") as f:
            f.setnchannels(1), f.setsampwidth(2), f.setframerate(16000)
            f.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
        items.append({"wav": p, "labels": ",".join(mids[c] for c in labs)})
    for name, part in (("train.json", items[:a.clips]), ("val.json", items[a.clips:])):
        with open(os.path.join(a.out, name), "w") as f:
            json.dump({"data": part}, f)
    print(f"{a.out}: {a.clips} train + {a.val_clips} val clips of {a.seconds}s, {a.classes} classes")


if __name__ == "__main__":
    main()
