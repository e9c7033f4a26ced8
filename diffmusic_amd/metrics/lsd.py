"""Log-spectral distance (reference: diffmusic/metrics/lsd.py:5-40, which calls librosa.stft with its defaults).

librosa is not a dependency here; the STFT it computes is restated: periodic Hann window of n_fft samples,
hop_length stride, center=True with ZERO padding of n_fft // 2 on both sides (librosa >= 0.10 default
pad_mode="constant"), one-sided rfft, 1 + L // hop frames."""
import numpy as np


def _stft_mag(x, n_fft, hop):
    x = np.asarray(x, dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="constant")
    n_frames = 1 + (xp.shape[-1] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)          # periodic Hann (scipy get_window fftbins=True)
    frames = xp[..., idx] * window
    return np.abs(np.fft.rfft(frames, axis=-1)).swapaxes(-1, -2)                  # (..., 1 + n_fft // 2, n_frames)


class LogSpectralDistance:
    def __init__(self, sample_rate=16000, n_fft=1024, hop_length=160, eps=1e-10):
        self.n_fft, self.hop_length, self.eps = n_fft, hop_length, eps

    def score(self, audio_background, audio_eval, output_mean=True):
        """(B, L) arrays -> mean over clips (or the (B,) vector) of mean_t sqrt(mean_f (log10|X| - log10|Y|)^2)."""
        ref = np.asarray(audio_background)
        est = np.nan_to_num(np.asarray(audio_eval), nan=0.0, posinf=1.0, neginf=-1.0)
        lr = np.log10(_stft_mag(ref, self.n_fft, self.hop_length) + self.eps)
        le = np.log10(_stft_mag(est, self.n_fft, self.hop_length) + self.eps)
        per_frame = np.sqrt(np.mean((lr - le) ** 2, axis=1))
        per_clip = per_frame.mean(axis=1)
        return per_clip.mean() if output_mean else per_clip
