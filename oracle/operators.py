"""Measurement operators restated from diffmusic/inverse_problem/operator.py and noise.py.
Devices follow the input tensor instead of the reference's hard-coded .to("cuda")
(operator.py:33,83,149,191,226)."""
import torch
import torch.nn.functional as F
from . import audio


class GaussianNoise:                                  # noise.py:13-18
    def __init__(self, sigma):
        self.sigma = sigma

    def __call__(self, data):
        return data + torch.randn_like(data) * self.sigma


def get_noiser(name, sigma):                          # inverse_problem/__init__.py:4-11
    if name == "gaussian":
        return GaussianNoise(sigma)
    raise ValueError(f"Unknown noise: {name}")


class BaseOperator:                                   # operator.py:6-14
    def transform(self, data, *a, **k):
        raise NotImplementedError

    def inverse_transform(self, mel_spectrogram, vocoder):   # operator.py:38-42 (six copies)
        if mel_spectrogram.dim() == 4:
            mel_spectrogram = mel_spectrogram.squeeze(1)
        return vocoder(mel_spectrogram)

    def forward(self, data, **k):
        raise NotImplementedError


class IdentityOperator(BaseOperator):                 # operator.py:17-45
    def __init__(self, sample_rate):
        self.wav2mel = audio.Wav2Mel(sample_rate)

    def transform(self, a):
        return torch.clamp(self.wav2mel(a), min=-80, max=80)

    def forward(self, data, **k):
        return data


class MusicInpaintingOperator(BaseOperator):          # operator.py:48-133
    def __init__(self, audio_length_in_s, sample_rate, mask_type, start_inpainting_s,
                 end_inpainting_s, mask_percentage, mask_duration_s, interval_s, noiser=None):
        self.audio_length_in_s, self.sample_rate, self.mask_type = audio_length_in_s, sample_rate, mask_type
        self.start_inpainting_s, self.end_inpainting_s = start_inpainting_s, end_inpainting_s
        self.mask_percentage, self.interval_s, self.mask_duration_s = mask_percentage, interval_s, mask_duration_s
        self.mask = self.generate_mask()
        self.wav2mel = audio.Wav2Mel(sample_rate)
        self.noiser = noiser

    def generate_mask(self):                          # operator.py:87-121
        mask = torch.ones([1, self.audio_length_in_s * self.sample_rate])
        sr = self.sample_rate
        if self.mask_type == "box":
            if self.start_inpainting_s is not None and self.end_inpainting_s is not None:
                mask[:, int(self.start_inpainting_s * sr): int(self.end_inpainting_s * sr)] = 0.
        elif self.mask_type == "random":
            total = self.audio_length_in_s * sr
            mask_samples = int(self.mask_percentage * total)
            dur = int(self.mask_duration_s * sr)
            for _ in range(max(1, mask_samples // dur)):
                start = torch.randint(0, mask.shape[1] - dur, (1,))
                mask[:, start:start + dur] = 0.
        elif self.mask_type == "periodic":
            interval, dur = int(self.interval_s * sr), int(self.mask_duration_s * sr)
            for start in range(0, mask.shape[1], interval):
                mask[:, start:min(start + dur, mask.shape[1])] = 0.
        return mask

    def transform(self, a):                           # operator.py:123-124 (no clamp)
        return self.wav2mel(a)

    def forward(self, data, **k):
        return self.noiser(data * self.mask.to(data.device))


class PhaseRetrievalOperator(BaseOperator):           # operator.py:136-171
    def __init__(self, n_fft=1024, hop_length=160, win_length=1024, noiser=None):
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.fb = audio.melscale_fbanks(1024 // 2 + 1, 0.0, 8000.0, 64, 16000)
        self.noiser = noiser

    def transform(self, magnitude):
        return torch.clamp(audio.mel_scale(magnitude.float(), self.fb), min=-80, max=80)

    def forward(self, data, **k):
        spec = torch.stft(data, n_fft=self.n_fft, hop_length=self.hop_length,
                          win_length=self.win_length, return_complex=True)   # window=None: rectangular
        return self.noiser(torch.abs(spec))


class SuperResolutionOperator(BaseOperator):          # operator.py:174-205
    def __init__(self, sample_rate, scale=10, noiser=None):
        self.orig, self.new = sample_rate, sample_rate // scale
        self.wav2mel = audio.Wav2Mel(16000)
        self.noiser = noiser

    def transform(self, a):
        return torch.clamp(self.wav2mel(a), min=-80, max=80)

    def forward(self, data, **k):
        return self.noiser(audio.resample(data.float(), self.orig, self.new))


class MusicDereverberationOperator(BaseOperator):     # operator.py:208-250
    def __init__(self, ir_length=800, decay_factor=0.85, noiser=None):
        self.ir_length, self.decay_factor = ir_length, decay_factor
        self.wav2mel = audio.Wav2Mel(16000)
        self.noiser = noiser

    def transform(self, a):
        return torch.clamp(self.wav2mel(a), min=-80, max=80)

    def generate_impulse_response(self, ir_length=800, decay_factor=0.85):   # operator.py:238-242
        ir = torch.randn(ir_length)
        ir = torch.cumsum(ir, dim=0) * decay_factor
        ir /= ir.abs().max()
        return ir.unsqueeze(0)

    def forward(self, data, ir=None, **k):
        """`ir` (extension): the reference draws a fresh IR from the global RNG on every call
        (operator.py:244-246); passing `ir` pins it for teacher-forced parity tests."""
        if ir is None:
            ir = self.generate_impulse_response(self.ir_length, self.decay_factor)
        ir = ir.to(data.device)
        out = F.conv1d(data.unsqueeze(1).float(), ir.unsqueeze(1), padding=ir.size(1) // 2).squeeze(1)
        return self.noiser(out)


class StyleGuidanceOperator(BaseOperator):            # operator.py:253-271; semantics defined by the build
    """transform(audio) = Gram(F) = F F^T / T of the CLAP (HTS-AT) audio-encoder token features F (B, C, T):
    16 -> 48 kHz sinc-hann resampling, CLAP log-mel (audio.clap_log_mel), transformers ClapAudioModel, all fp32 eager.
    **Parity unpinned**: the reference's `clap_model.get_gram_matrix` is undefined and run.py:213-214 raises for this task."""

    def __init__(self, sample_rate=16000, clap_model=None, noiser=None):
        self.sample_rate, self.noiser = sample_rate, noiser
        self.clap = getattr(clap_model, "audio_model", clap_model)
        self.fb = audio.mel_filter_bank_slaney()

    def forward(self, data, **k):
        return self.noiser(data) if self.noiser is not None else data

    def features(self, a):
        return audio.clap_log_mel(audio.resample(a.float(), self.sample_rate, 48000), self.fb)[:, None]

    def transform(self, a):
        f = self.clap(input_features=self.features(a), is_longer=None, return_dict=True).last_hidden_state.flatten(2)
        return torch.bmm(f, f.transpose(1, 2)) / f.shape[2]
