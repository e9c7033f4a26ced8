"""AudioLDM2 pipeline facade (reference: diffmusic/pipelines/plpeline_audioldm2.py:924-1254).  Same loop as MusicLDM
(the reference's two loops differ only in conditioning, :1147-1154, and in the default guidance_scale 3.5, :930);
the U-Net attends two contexts: `generated_prompt_embeds` (B, 8, 768) from GPT-2 and `prompt_embeds` (B, L, 1024) from
T5 with `attention_mask`.  Pass the embeddings directly, as the reference signature allows, or attach the wrapped
CLAP / T5 / projection / GPT-2 front end (`pipe.text_frontend = AudioLDM2PromptFrontEnd(...)`, prompt_audioldm2.py) and
call with `prompt=` (and `prompt_type="clap"` for the audio-conditioned ablation, :469-481)."""
import torch

from ..engine import UNET_AUDIOLDM2_DEFAULT
from .pipeline_musicldm import MusicLDMPipeline


class AudioLDM2Pipeline(MusicLDMPipeline):
    default_guidance_scale = 3.5
    unet_default_config = UNET_AUDIOLDM2_DEFAULT

    def _prepare_cond(self, prompt_embeds, negative_prompt_embeds, n_per, do_cfg, device, generated_prompt_embeds=None,
                      negative_generated_prompt_embeds=None, attention_mask=None, negative_attention_mask=None, **_):
        if generated_prompt_embeds is None:
            raise NotImplementedError("no prompt front end attached: pass prompt_embeds (B, L, 1024) + generated_prompt_embeds (B, 8, 768), "
                                      "or set pipe.text_frontend = AudioLDM2PromptFrontEnd(...) to use `prompt=`")

        def rep(x):
            return None if x is None else x.to(device=device, dtype=torch.float32).repeat_interleave(n_per, dim=0)
        pe, ge, am = rep(prompt_embeds), rep(generated_prompt_embeds), rep(attention_mask)
        if am is None:
            am = torch.ones(pe.shape[:2], device=device)
        if do_cfg:                                                   # plpeline_audioldm2.py:640-668: [negative | positive]
            npe = rep(negative_prompt_embeds) if negative_prompt_embeds is not None else pe
            nge = rep(negative_generated_prompt_embeds) if negative_generated_prompt_embeds is not None else ge
            nam = rep(negative_attention_mask) if negative_attention_mask is not None else am
            pe, ge, am = torch.cat([npe, pe]), torch.cat([nge, ge]), torch.cat([nam, am])
        return dict(class_labels=None, encoder_hidden_states=ge, encoder_hidden_states_1=pe, encoder_attention_mask_1=am)

    def _needs_negative_embeds(self):
        return False            # the unconditional branch is built per context in _prepare_cond (:640-668)

    def _optim_prompt_step(self, noise_pred, t, latents, cond, measurement, length, lr, supervised_space, extra):
        """plpeline_audioldm2.py:1161-1176: both contexts go through scheduler.optim_prompt and come back (unchanged)."""
        out = self.scheduler.optim_prompt(noise_pred, t, latents, measurement=measurement, original_waveform_length=length,
                                          vae=self.vae, vocoder=self.vocoder, encoder_hidden_states=cond["encoder_hidden_states"],
                                          encoder_hidden_states_1=cond["encoder_hidden_states_1"], optim_prompt_learning_rate=lr,
                                          supervised_space=supervised_space, **extra)
        return dict(cond, encoder_hidden_states=out.encoder_hidden_states, encoder_hidden_states_1=out.encoder_hidden_states_1)

    def __call__(self, prompt=None, transcription=None, audio_length_in_s=None, num_inference_steps=200, guidance_scale=3.5,
                 negative_prompt=None, num_waveforms_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, generated_prompt_embeds=None, negative_generated_prompt_embeds=None,
                 attention_mask=None, negative_attention_mask=None, max_new_tokens=None, **kw):
        if prompt_embeds is None and prompt is not None and getattr(self, "text_frontend", None) is not None:
            enc = self.text_frontend.encode(prompt, negative_prompt, guidance_scale > 1.0, max_new_tokens=max_new_tokens,
                                            prompt_type=kw.get("prompt_type"), measurement=kw.get("measurement"),
                                            transcription=transcription)                  # plpeline_audioldm2.py:1077-1090
            prompt_embeds, attention_mask, generated_prompt_embeds = (enc["prompt_embeds"], enc["attention_mask"],
                                                                      enc["generated_prompt_embeds"])
            if negative_prompt_embeds is None and "negative_prompt_embeds" in enc:
                negative_prompt_embeds, negative_attention_mask, negative_generated_prompt_embeds = (
                    enc["negative_prompt_embeds"], enc["negative_attention_mask"], enc["negative_generated_prompt_embeds"])
        self._extra_cond = dict(generated_prompt_embeds=generated_prompt_embeds,
                                negative_generated_prompt_embeds=negative_generated_prompt_embeds,
                                attention_mask=attention_mask, negative_attention_mask=negative_attention_mask)
        try:
            return super().__call__(prompt=prompt, audio_length_in_s=audio_length_in_s, num_inference_steps=num_inference_steps,
                                    guidance_scale=guidance_scale, negative_prompt=negative_prompt,
                                    num_waveforms_per_prompt=num_waveforms_per_prompt, eta=eta, generator=generator, latents=latents,
                                    prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, **kw)
        finally:
            self._extra_cond = {}
