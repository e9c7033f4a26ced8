"""Log-spectral distance (reference: diffmusic/metrics/lsd.py:5-40, which calls librosa.stft with its defaults).

librosa is not a dependency here; the STFT it computes is restated: periodic Hann window of n_fft samples,
hop_length stride, center=True with ZERO padding of n_fft // 2 on both sides (librosa >= 0.10 default
pad_mode="constant"), one-sided rfft, 1 + L // hop frames.

Two legs with the same definition: numpy float64 on the host (the reference's form: lists / arrays in), and -- for CUDA tensors -- a
float64 STFT on the device (`torch.stft` in fp64 + device-side reductions), so a batch that is already on the GPU after generation is
scored there and agrees with the host leg to ~1e-9, silent passages included.  `device_precision="fp32"` selects the HIP
STFT-magnitude kernel instead (csrc/stft_mel.hip through SpectralFrontend.stft_mag): faster, but its fp32 noise floor (~1e-6 of the
clip's peak) replaces magnitudes below it, and log10(|X| + 1e-10) of a masked gap or of a band-limited reference then differs from the
float64 definition by whole units -- use it only for clips without near-silent bins (agreement elsewhere: 2e-4 relative)."""
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
    def __init__(self, sample_rate=16000, n_fft=1024, hop_length=160, eps=1e-10, device_precision="fp64"):
        if device_precision not in ("fp64", "fp32"):
            raise ValueError("device_precision: 'fp64' (the host leg's definition) or 'fp32' (HIP STFT kernel)")
        self.sample_rate, self.n_fft, self.hop_length, self.eps, self.device_precision = sample_rate, n_fft, hop_length, eps, device_precision

    def _mag_gpu64(self, x):
        """(B, L) cuda -> (B, 1 + n_fft/2, 1 + L // hop) float64 magnitudes: periodic Hann, zero centre padding (librosa's defaults)."""
        import torch
        win = torch.hann_window(self.n_fft, periodic=True, dtype=torch.float64, device=x.device)
        return torch.stft(x.double(), self.n_fft, hop_length=self.hop_length, window=win, center=True, pad_mode="constant",
                          return_complex=True).abs()

    def _mag_gpu(self, x):
        """(B, L) fp32 cuda -> (B, 1 + n_fft/2, 1 + L // hop) magnitudes with ZERO centre padding.  The kernel pads by reflection
        (torch.stft): the clip is embedded in zeros, P >= n_fft/2 + 1 of them in front with P a multiple of the hop, so that the
        reflected samples are zeros and frame i of the zero-padded clip is frame i + P / hop of the embedded one."""
        import torch
        from ..inverse_problem.operator import SpectralFrontend
        if getattr(self, "_fe", None) is None:
            self._fe = SpectralFrontend(self.sample_rate, self.n_fft, self.hop_length, 64, "hann")
        B, L = x.shape
        half = self.n_fft // 2
        P = -(-(half + 1) // self.hop_length) * self.hop_length
        xp = torch.zeros(B, P + L + P, dtype=torch.float32, device=x.device)
        xp[:, P:P + L] = x
        k = P // self.hop_length
        return self._fe.stft_mag(xp, xp.shape[1])[:, :, k:k + 1 + L // self.hop_length]

    def _score_gpu(self, ref, est, output_mean):
        import torch
        ref = ref.float().reshape(-1, ref.shape[-1]).contiguous()
        est = torch.nan_to_num(est.float().reshape(-1, est.shape[-1]), nan=0.0, posinf=1.0, neginf=-1.0).contiguous()
        mag = self._mag_gpu64 if self.device_precision == "fp64" else self._mag_gpu
        lr = torch.log10(mag(ref) + self.eps)
        le = torch.log10(mag(est) + self.eps)
        per_clip = ((lr - le) ** 2).mean(dim=1).sqrt().mean(dim=1)
        return per_clip.mean() if output_mean else per_clip

    def score(self, audio_background, audio_eval, output_mean=True):
        """(B, L) arrays -> mean over clips (or the (B,) vector) of mean_t sqrt(mean_f (log10|X| - log10|Y|)^2).
        CUDA tensors in -> computed on the GPU, CUDA tensor out."""
        if getattr(audio_background, "is_cuda", False) and getattr(audio_eval, "is_cuda", False):
            return self._score_gpu(audio_background, audio_eval, output_mean)
        ref = np.asarray(audio_background)
        est = np.nan_to_num(np.asarray(audio_eval), nan=0.0, posinf=1.0, neginf=-1.0)
        lr = np.log10(_stft_mag(ref, self.n_fft, self.hop_length) + self.eps)
        le = np.log10(_stft_mag(est, self.n_fft, self.hop_length) + self.eps)
        per_frame = np.sqrt(np.mean((lr - le) ** 2, axis=1))
        per_clip = per_frame.mean(axis=1)
        return per_clip.mean() if output_mean else per_clip
