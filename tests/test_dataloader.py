"""CPU: the WAV dataset crop / resample loader (reference: diffmusic/data/dataloader.py:47-89)."""
import os
import struct
import wave

import numpy as np
import pytest
import torch

from diffmusic_amd.data import WAVDataset, get_dataloader, get_dataset


def _write(path, x, sr, width=2):
    x = np.atleast_2d(x)
    with wave.open(path, "wb") as w:
        w.setnchannels(x.shape[0]); w.setsampwidth(width); w.setframerate(sr)
        if width == 2:
            w.writeframes((np.clip(x.T, -1, 1) * 32767).astype("<i2").tobytes())
        else:
            v = (np.clip(x.T, -1, 1) * 8388607).astype(np.int32).reshape(-1)
            w.writeframes(b"".join(struct.pack("<i", int(s))[:3] for s in v))


def test_wav_dataset_crop_mono_resample(tmp_path):
    from oracle.audio import resample
    sr = 16000
    t = np.arange(3 * sr) / sr
    a = 0.5 * np.sin(2 * np.pi * 440 * t)
    _write(str(tmp_path / "b_16k_stereo.wav"), np.stack([a, 0.5 * a]), sr)
    t44 = np.arange(3 * 44100) / 44100
    _write(str(tmp_path / "a_44k.wav"), 0.5 * np.sin(2 * np.pi * 440 * t44), 44100, width=3)
    os.makedirs(tmp_path / "sub")
    _write(str(tmp_path / "sub" / "c.wav"), a, sr)
    ds = get_dataset(name="moises", type="wav", root=str(tmp_path), sample_rate=sr, audio_length_in_s=1, start_s=1, end_s=2)
    assert isinstance(ds, WAVDataset) and len(ds) == 3
    names = [ds[i][1] for i in range(3)]
    assert names == ["a_44k.wav", "b_16k_stereo.wav", "c.wav"]                    # sorted recursive glob
    w44, _ = ds[0]
    wst, _ = ds[1]
    assert w44.shape == wst.shape == (sr,)
    assert np.allclose(wst.numpy(), 0.75 * a[sr:2 * sr], atol=2e-4)                # stereo -> mean of channels, crop [1 s, 2 s)
    ref = resample(torch.from_numpy((np.clip(0.5 * np.sin(2 * np.pi * 440 * t44), -1, 1) * 8388607).astype(np.int32).astype(np.float32)
                                    / 8388608.0)[None], 44100, sr)[0, sr:2 * sr]
    assert torch.allclose(w44, ref, atol=1e-5)                                     # torchaudio-style sinc resampling (oracle.audio.resample)
    assert float((w44 - torch.from_numpy(a[sr:2 * sr]).float()).abs().max()) < 5e-3   # and it is the same tone
    dl = get_dataloader(ds, batch_size=3, num_workers=0, train=False)
    batch, fn = next(iter(dl))
    assert batch.shape == (3, sr) and list(fn) == names
    with pytest.raises(NameError):
        get_dataset(name="x", type="flac", root=str(tmp_path))
    with pytest.raises(AssertionError):
        WAVDataset(str(tmp_path / "sub" / "nothing"), sr, 1)


class _FakeSegment:
    """Stand-in for pydub.AudioSegment (pydub / ffmpeg are absent here): int16 samples, interleaved channels; `set_frame_rate` decimates /
    repeats by nearest index and `set_channels(1)` averages -- enough to observe the ORDER and the scaling the dataset applies."""
    array_type = "h"
    calls = []

    def __init__(self, samples, frame_rate, channels):
        self._s, self.frame_rate, self.channels = np.asarray(samples, dtype=np.int16), frame_rate, channels

    @classmethod
    def from_file(cls, path, format):
        assert format == "mp3"
        cls.calls.append(("from_file", os.path.basename(path)))
        sr = 32000 if "32k" in path else 16000
        n = 2 * sr
        left = (np.arange(n) % 1000 * 30).astype(np.int16)
        return cls(np.stack([left, -left], axis=1).reshape(-1) if "stereo" in path else left, sr, 2 if "stereo" in path else 1)

    def set_frame_rate(self, sr):
        type(self).calls.append(("set_frame_rate", sr, self.channels))
        x = self._s.reshape(-1, self.channels)
        idx = (np.arange(int(len(x) * sr / self.frame_rate)) * self.frame_rate / sr).astype(int)
        return _FakeSegment(x[idx].reshape(-1), sr, self.channels)

    def set_channels(self, n):
        assert n == 1
        type(self).calls.append(("set_channels", n))
        return _FakeSegment(self._s.reshape(-1, self.channels).astype(np.int32).mean(axis=1).astype(np.int16), self.frame_rate, 1)

    def get_array_of_samples(self):
        return self._s


def test_mp3_dataset_follows_the_reference_order(tmp_path, monkeypatch):
    """Reference: diffmusic/data/dataloader.py:92-145 -- resample, then mono, / iinfo.max, crop (end_s <= 0: to the end), then transform."""
    import sys
    import types
    from diffmusic_amd.data import MP3Dataset
    for n in ("b_32k_stereo.mp3", "a.mp3"):
        (tmp_path / n).write_bytes(b"\0")
    ds = get_dataset(name="x", type="mp3", root=str(tmp_path), sample_rate=16000, audio_length_in_s=1, start_s=0.5, end_s=0)
    assert isinstance(ds, MP3Dataset) and len(ds) == 2
    monkeypatch.setitem(sys.modules, "pydub", None)                 # import pydub -> ImportError: the loud failure
    with pytest.raises(ImportError, match="pydub"):
        ds[0]
    fake = types.ModuleType("pydub")
    fake.AudioSegment = _FakeSegment
    monkeypatch.setitem(sys.modules, "pydub", fake)
    _FakeSegment.calls.clear()
    w, name = ds[0]
    assert name == "a.mp3" and w.dtype == torch.float32 and w.shape == (2 * 16000 - 8000,)
    assert torch.allclose(w, torch.from_numpy(((np.arange(32000) % 1000 * 30).astype(np.float32) / 32767.0)[8000:]))
    assert _FakeSegment.calls == [("from_file", "a.mp3")]           # already 16 kHz mono: neither conversion is called
    _FakeSegment.calls.clear()
    ds.end_s, ds.transforms = 1.0, (lambda t: 2.0 * t)
    w, name = ds[1]
    assert name == "b_32k_stereo.mp3" and w.shape == (8000,)
    assert [c[0] for c in _FakeSegment.calls] == ["from_file", "set_frame_rate", "set_channels"]
    assert _FakeSegment.calls[1] == ("set_frame_rate", 16000, 2)     # still stereo when resampled
    assert float(w.abs().max()) == 0.0                               # L = -R: the mono mix is silence, doubled by the transform
    with pytest.raises(AssertionError):
        MP3Dataset(str(tmp_path / "none"), 16000, 1)
