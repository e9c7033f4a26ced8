"""Case table shared by the golden generator (build container) and the tests (anywhere)."""
SCHED_CFG = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195,
                 beta_schedule="scaled_linear", trained_betas=None, clip_sample=False,
                 set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                 sample_max_value=1.0, timestep_spacing="leading", rescale_betas_zero_snr=False)

SR = 16000
H, W = 10, 4                      # latent (1,8,10,4) -> mel (1,1,40,16) -> wav 40*160 = 6400 samples
L = 6400


CASES = [("ddim", "music_inpainting", 0.0, 0.0, 50, "mel_spectrogram"),
         ("dps", "music_inpainting", 0.0, 5e-4, 200, "mel_spectrogram"),
         ("dps", "music_inpainting", 0.0, 5e-4, 200, "wav_form"),
         ("dps", "super_resolution", 0.0, 5e-4, 200, "mel_spectrogram"),
         ("mpgd", "music_inpainting", 0.0, 5e-3, 200, "mel_spectrogram"),
         ("mpgd", "phase_retrieval", 0.0, 5e-3, 200, "mel_spectrogram"),
         ("dsg", "music_inpainting", 1.0, 0.08, 200, "mel_spectrogram"),
         ("dsg", "phase_retrieval", 1.0, 0.08, 200, "mel_spectrogram"),
         ("diffmusic", "music_inpainting", 1.0, 0.08, 200, "mel_spectrogram"),
         ("diffmusic", "music_inpainting", 1.0, 0.9999, 200, "mel_spectrogram"),
         # eta > 0 on the deterministic samplers: the parent DDIM step consumes one draw of the generator before the scheduler's own
         # variance noise (scheduling_dps.py:166-193, scheduling_mpgd.py:164-173,206-217).  Appended: earlier cases keep their seeds.
         ("dps", "music_inpainting", 0.5, 5e-4, 200, "mel_spectrogram"),
         ("mpgd", "music_inpainting", 0.5, 5e-3, 200, "mel_spectrogram")]


