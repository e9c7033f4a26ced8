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
