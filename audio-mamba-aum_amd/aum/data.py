"""Waveform datasets for the GPU frontend.  Reads the reference's data files unchanged — the `{"data": [{"wav": ...,
"labels": "mid1,mid2"}]}` json lists and the `index,mid,display_name` label csv of /root/reference/src/dataloader.py
("DL":13-21, 59-63) — but hands WAVEFORMS to the training loop; mel, SpecAug and normalisation then run on the
GPU (aum.frontend / aum.augment) instead of in CPU workers.  Mix-up stays here because its partner is drawn from
the whole dataset (DL:158-184), and is a cheap axpy on the waveform.

Decoding: PCM `.wav` through the standard library and `.npy` float arrays.  torchaudio is not in this image; other
containers raise."""
import csv
import json
import random
import wave as _wave

import numpy as np
import torch
from torch.utils.data import Dataset

from .augment import mixup_waveforms


def make_index_dict(label_csv):
    with open(label_csv, "r") as f:
        return {row["mid"]: int(row["index"]) for row in csv.DictReader(f)}


def make_name_dict(label_csv):
    with open(label_csv, "r") as f:
        return {row["index"]: row["display_name"] for row in csv.DictReader(f)}


def read_audio(path):
    """-> (mono float32 waveform in [-1, 1), sample_rate); channel 0 of multi-channel files is NOT special-cased by
    the reference (kaldi.fbank takes channel 0 by default), so neither here."""
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32).reshape(-1), None
    if not path.lower().endswith(".wav"):
        raise ValueError(f"{path}: only PCM .wav and .npy waveforms can be decoded in this build")
    with _wave.open(path, "rb") as f:
        nch, width, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return x.reshape(-1, nch)[:, 0].copy(), sr


class WaveformDataset(Dataset):
    """item -> (waveform zero-padded/cut to max_samples, n_valid_samples, label vector, path)."""
    MAX_RETRIES = 20

    def __init__(self, dataset_json_file, label_csv, max_samples, mixup=0.0, sample_rate=16000):
        with open(dataset_json_file, "r") as fp:
            self.data = json.load(fp)["data"]
        self.index_dict = make_index_dict(label_csv)
        self.label_num = len(self.index_dict)
        self.max_samples, self.mixup, self.sample_rate = max_samples, mixup, sample_rate

    def __len__(self):
        return len(self.data)

    def _load(self, path):
        x, sr = read_audio(path)
        if sr is not None and sr != self.sample_rate:
            raise ValueError(f"{path}: sample rate {sr} != configured {self.sample_rate}")
        return x - x.mean()                                            # DL:101

    def _labels(self, datum, weight, out):
        for s in datum["labels"].split(","):
            out[self.index_dict[s]] += weight

    def __getitem__(self, index):
        labels = np.zeros(self.label_num, np.float32)
        tries = 0
        while True:                                                    # DL:160-171, 189-199: retry another clip
            datum = self.data[index]
            try:
                w = self._load(datum["wav"])
                if random.random() < self.mixup:
                    other = self.data[random.randint(0, len(self.data) - 1)]
                    w, lam = mixup_waveforms(w, self._load(other["wav"]))
                    self._labels(datum, lam, labels)
                    self._labels(other, 1.0 - lam, labels)
                else:
                    self._labels(datum, 1.0, labels)
                    labels = np.minimum(labels, 1.0)                   # DL:203 assigns, duplicates do not add
                break
            except (OSError, EOFError, ValueError, _wave.Error) as e:
                # ValueError: unsupported container / sample width / sample rate (read_audio); the reference retries another
                # clip on any load failure (DL:160-171).  Bounded, so a list of unreadable files fails instead of spinning.
                tries += 1
                print(f"dataloading failed for {datum['wav']} ({e}), retrying ({tries}/{self.MAX_RETRIES})...")
                if tries >= self.MAX_RETRIES:
                    raise RuntimeError(f"{self.MAX_RETRIES} consecutive clips could not be loaded; last: {datum['wav']}") from e
                index = random.randint(0, len(self.data) - 1)
                labels[:] = 0
        n = min(len(w), self.max_samples)
        buf = np.zeros(self.max_samples, np.float32)
        buf[:n] = w[:n]
        return torch.from_numpy(buf), n, torch.from_numpy(labels), datum["wav"]
