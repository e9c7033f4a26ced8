"""Dataset crop / resample loader (reference: diffmusic/data/dataloader.py:15-89): the same
registry (`register_dataset`, `get_dataset(name, type, root, **kw)`, `get_dataloader`) and the `wav` dataset -- mono mix-down,
resampling to `sample_rate`, optional transform, crop to [start_s, end_s), item = (waveform, file name).

torchaudio is absent from this image: WAV files are decoded with the standard library (`wave`: 8/16/24/32-bit PCM)
or scipy (float WAVs), and resampling uses the same sinc-hann polyphase kernel as the measurement operators
(torchaudio `Resample` semantics): on the host by default (the loader runs before the hot loop, in
DataLoader workers), or -- `device="cuda"` -- through the HIP FIR kernel (csrc/fir.hip), with the item left on the GPU.
The `mp3` dataset (reference: dataloader.py:92-145) decodes through pydub / ffmpeg exactly as the reference does; pydub is
imported when the first item is read, so the class is registered everywhere and fails loudly where the decoder is missing."""
import os
import wave as _wave
from glob import glob

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from ..pipelines.prompt_audioldm2 import resample_to

_REGISTRY = {}                    # dataset type ("wav") -> class


def register_dataset(name):
    """Class decorator: make `cls` available to `get_dataset(type=name)`; a type can be registered once."""
    def bind(cls):
        if name in _REGISTRY:
            raise NameError(f"dataset type {name!r} is already registered ({_REGISTRY[name].__name__})")
        _REGISTRY[name] = cls
        return cls
    return bind


def get_dataset(name, type, root, **kwargs):
    """`name` is the corpus label of the reference's configs (unused by the loader itself), `type` selects the registered class."""
    try:
        cls = _REGISTRY[type]
    except KeyError:
        raise NameError(f"dataset type {type!r} is not registered (known: {sorted(_REGISTRY)})") from None
    return cls(root=root, **kwargs)


def get_dataloader(dataset, batch_size, num_workers, train):
    # evaluation keeps file order and the ragged last batch; training shuffles and drops it
    return DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, shuffle=bool(train), drop_last=bool(train))


def load_wav(path):
    """-> (float32 tensor (channels, time) in [-1, 1], sample_rate), like torchaudio.load(path)."""
    try:
        with _wave.open(path, "rb") as w:
            ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
        if width == 1:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif width == 3:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
        elif width == 4:
            x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"unsupported sample width {width}")
        x = x.reshape(-1, ch).T
    except _wave.Error:                                   # IEEE-float WAV: the wave module refuses it
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        data = np.asarray(data)
        if data.dtype.kind == "i":
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
        elif data.dtype.kind == "u":
            data = (data.astype(np.float32) - 128.0) / 128.0
        x = np.atleast_2d(data.astype(np.float32).T if data.ndim == 2 else data.astype(np.float32))
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)


@register_dataset(name="wav")
class WAVDataset(Dataset):
    """All `*.wav` files below `root` (sorted), each as (mono waveform cropped to [start_s, end_s) at `sample_rate`, file name).
    Reference: diffmusic/data/dataloader.py:47-89."""

    def __init__(self, root, sample_rate, audio_length_in_s, start_s=0, end_s=0, transforms=None, device=None):
        """device: None = host tensors (reference behaviour); "cuda" = mix-down, resampling (HIP FIR) and crop on the GPU, items stay
        there (use num_workers=0: a CUDA context does not survive the fork of a DataLoader worker)."""
        self.root = root
        self.sample_rate = sample_rate
        self.audio_length_in_s = audio_length_in_s
        self.start_s, self.end_s = start_s, end_s
        self.transforms = transforms
        self.device = torch.device(device) if device is not None else None
        self.fpaths = sorted(glob(os.path.join(root, "**", "*.wav"), recursive=True))
        if not self.fpaths:
            raise AssertionError(f"no .wav files below {root!r}")

    def __len__(self):
        return len(self.fpaths)

    def _crop(self):
        return slice(int(self.start_s * self.sample_rate), int(self.end_s * self.sample_rate))

    def __getitem__(self, index):
        path = self.fpaths[index]
        audio, rate = load_wav(path)                        # (channels, time)
        if self.device is not None:
            audio = audio.to(self.device)
        mono = audio if audio.shape[0] == 1 else audio.mean(dim=0, keepdim=True)
        if rate != self.sample_rate:
            mono = resample_to(mono, rate, self.sample_rate)
        if self.transforms is not None:
            mono = self.transforms(mono)
        return mono[0][self._crop()], os.path.basename(path)


@register_dataset(name="mp3")
class MP3Dataset(Dataset):
    """All `*.mp3` files below `root` (sorted), decoded with pydub (ffmpeg).  Reference: diffmusic/data/dataloader.py:92-145 -- and its
    order of operations, which differs from the WAV dataset: pydub resamples (`set_frame_rate`) BEFORE the mono mix-down
    (`set_channels(1)`), samples are scaled by 1 / max(int type) (32767 for 16-bit, not 32768), `end_s <= 0` means "to the end of the
    file", and the transform sees the cropped clip (the WAV dataset transforms before cropping)."""

    def __init__(self, root, sample_rate, audio_length_in_s, start_s=0, end_s=0, transforms=None):
        self.root = root
        self.sample_rate = sample_rate
        self.audio_length_in_s = audio_length_in_s
        self.start_s, self.end_s = start_s, end_s
        self.transforms = transforms
        self.fpaths = sorted(glob(os.path.join(root, "**", "*.mp3"), recursive=True))
        if not self.fpaths:
            raise AssertionError(f"no .mp3 files below {root!r}")

    def __len__(self):
        return len(self.fpaths)

    def __getitem__(self, index):
        try:
            from pydub import AudioSegment
        except ImportError as e:
            raise ImportError("the mp3 dataset decodes with pydub (+ ffmpeg), which is not installed; convert the corpus to .wav "
                              "and use type='wav'") from e
        path = self.fpaths[index]
        seg = AudioSegment.from_file(path, format="mp3")
        if seg.frame_rate != self.sample_rate:
            seg = seg.set_frame_rate(self.sample_rate)
        if seg.channels > 1:
            seg = seg.set_channels(1)
        x = np.array(seg.get_array_of_samples()).astype(np.float32)
        x /= np.iinfo(seg.array_type).max
        lo = int(self.start_s * self.sample_rate)
        hi = int(self.end_s * self.sample_rate) if self.end_s > 0 else len(x)
        clip = torch.from_numpy(x[lo:hi].copy())
        if self.transforms is not None:
            clip = torch.as_tensor(self.transforms(clip))
        return clip, os.path.basename(path)
