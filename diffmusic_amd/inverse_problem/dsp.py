"""Host-side DSP tables (built once, uploaded to the HIP library): torchaudio-compatible HTK mel
filterbank (norm=None) and the sinc-hann polyphase resampling kernel."""
import math
import numpy as np


def melscale_fbanks(n_freqs=513, f_min=0.0, f_max=8000.0, n_mels=64, sample_rate=16000):
    """(n_freqs, n_mels) float32 triangular filters, HTK mel scale, no area normalisation."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2, dtype=np.float32)
    f_pts = (700.0 * (np.power(np.float32(10.0), m_pts / np.float32(2595.0)) - 1.0)).astype(np.float32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Returns (kernel (new, taps) float32, width, orig, new) after gcd reduction."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * (base / orig)
    return kern.astype(np.float32), width, orig, new
