"""torchaudio transform semantics restated in plain torch (reference call sites:
diffmusic/inverse_problem/operator.py:23-33 wav2mel, :143-147 MelScale, :180 Resample).
torchaudio is absent from this image: parity with it is unpinned; the mel filterbank is
cross-checked against transformers.audio_utils.mel_filter_bank and the STFT is torch.stft
itself (tests/test_oracle_audio.py)."""
import math
import torch
import torch.nn.functional as F


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs=513, f_min=0.0, f_max=8000.0, n_mels=64, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = hz_to_mel_htk(f_min), hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)       # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def power_spectrogram(wav, n_fft=1024, hop=160, win_length=1024, power=2.0):
    """torchaudio Spectrogram: periodic hann, center=True reflect pad, onesided. (B,L)->(B,F,T)."""
    window = torch.hann_window(win_length, periodic=True, dtype=wav.dtype, device=wav.device)
    spec = torch.stft(wav, n_fft=n_fft, hop_length=hop, win_length=win_length, window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    mag = spec.abs()
    return mag ** power if power != 1.0 else mag


def mel_scale(spec, fb):
    """torchaudio MelScale: (B, F, T) x fb (F, M) -> (B, M, T)."""
    return torch.matmul(spec.transpose(-1, -2), fb.to(spec)).transpose(-1, -2)


def amplitude_to_db_power(x, amin=1e-10):
    """AmplitudeToDB('power', top_db=None): 10*log10(clamp(x, amin)) - 10*log10(max(amin, 1))."""
    return 10.0 * torch.log10(torch.clamp(x, min=amin))


class Wav2Mel:
    """Sequential(MelSpectrogram(sr,1024,160,1024,n_mels=64,power=2), AmplitudeToDB('power'))."""

    def __init__(self, sample_rate=16000, n_fft=1024, hop_length=160, win_length=1024, n_mels=64,
                 power=2.0):
        self.n_fft, self.hop, self.win, self.power = n_fft, hop_length, win_length, power
        self.fb = melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)

    def __call__(self, wav):
        p = power_spectrogram(wav, self.n_fft, self.hop, self.win, self.power)
        return amplitude_to_db_power(mel_scale(p, self.fb))


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = t * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    kern = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kern = kern * window * scale
    return kern.to(torch.float32), width, orig, new        # kern: (new, 1, 2*width+orig)


def resample(wav, orig_freq, new_freq):
    """torchaudio Resample.forward == _apply_sinc_resample_kernel.  (B, L) -> (B, ceil(new*L/orig))."""
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    if orig == new:
        return wav
    B, L = wav.shape
    x = F.pad(wav[:, None], (width, width + orig))
    y = F.conv1d(x, kern.to(wav), stride=orig)                 # (B, new, n)
    y = y.transpose(1, 2).reshape(B, -1)
    target = int(math.ceil(new * L / orig))
    return y[:, :target]


def mel_filter_bank_slaney(n_freqs=513, n_mels=64, f_min=0.0, f_max=14000.0, sample_rate=48000):
    """librosa.filters.mel(htk=False, norm='slaney') == transformers ClapFeatureExtractor.mel_filters_slaney -> (n_freqs, n_mels).
    Slaney mel scale: linear below 1 kHz (3 mel per 200 Hz), logarithmic above; area-normalised triangles."""
    def hz_to_mel(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        lin = 3.0 * f / 200.0
        log = 15.0 + torch.log(f.clamp_min(1e-10) / 1000.0) * (27.0 / math.log(6.4))
        return torch.where(f >= 1000.0, log, lin)

    def mel_to_hz(m):
        lin = 200.0 * m / 3.0
        log = 1000.0 * torch.exp((math.log(6.4) / 27.0) * (m - 15.0))
        return torch.where(m >= 15.0, log, lin)
    fft_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(float(hz_to_mel(f_min)), float(hz_to_mel(f_max)), n_mels + 2, dtype=torch.float64)
    f_pts = mel_to_hz(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return (fb * enorm[None, :]).to(torch.float32)


def clap_log_mel(wav48, fb=None):
    """transformers ClapFeatureExtractor._np_extract_fbank_features (truncation != 'fusion'): periodic hann(1024), hop 480,
    centre / reflect, power spectrogram -> slaney mel -> 10 log10(max(., 1e-10)).  (B, L) -> (B, frames, 64)."""
    fb = mel_filter_bank_slaney() if fb is None else fb
    window = torch.hann_window(1024, periodic=True, dtype=wav48.dtype, device=wav48.device)
    spec = torch.stft(wav48, n_fft=1024, hop_length=480, win_length=1024, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True).abs() ** 2                       # (B, 513, T)
    mel = torch.matmul(spec.transpose(1, 2), fb.to(spec))                                                        # (B, T, 64)
    return 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
