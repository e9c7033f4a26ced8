import torch
from oracle import audio


class MelSpectrogram(torch.nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, hop_length=None, win_length=None, n_mels=128, power=2.0):
        super().__init__()
        self.a = (n_fft, hop_length, win_length, power)
        self.fb = audio.melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)

    def forward(self, x):
        n_fft, hop, win, power = self.a
        return audio.mel_scale(audio.power_spectrogram(x, n_fft, hop, win, power), self.fb)


class AmplitudeToDB(torch.nn.Module):
    def __init__(self, stype="power", top_db=None):
        super().__init__()
        assert stype == "power" and top_db is None

    def forward(self, x):
        return audio.amplitude_to_db_power(x)


class MelScale(torch.nn.Module):
    def __init__(self, n_mels=128, sample_rate=16000, f_min=0.0, f_max=None, n_stft=201):
        super().__init__()
        self.fb = audio.melscale_fbanks(n_stft, f_min, f_max or float(sample_rate // 2), n_mels, sample_rate)

    def forward(self, x):
        return audio.mel_scale(x, self.fb)


class Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000):
        super().__init__()
        self.o, self.n = orig_freq, new_freq

    def forward(self, x):
        return audio.resample(x, self.o, self.n)
