"""Prompt-conditioning front end for AudioLDM2: CLAP text (or audio) tower + T5 encoder ->
projection model -> GPT-2 generating `max_new_tokens` (8) hidden states, following `AudioLDM2Pipeline.encode_prompt` and
`generate_language_model` (diffmusic/pipelines/plpeline_audioldm2.py:280-668).  The encoders and the language model are
`transformers` modules that are WRAPPED, not re-implemented; the small projection model (diffusers
`AudioLDM2ProjectionModel`, absent from this image) is restated below with the upstream parameter names so its checkpoint
loads 1:1.  Runs once per call, outside the hot loop; the engine only ever sees the resulting tensors:

    prompt_embeds            (B, L, 1024)  T5 last hidden state       -> U-Net context 1 (+ attention_mask (B, L))
    generated_prompt_embeds  (B, 8, 768)   GPT-2 generated states     -> U-Net context 0

    front = AudioLDM2PromptFrontEnd.from_pretrained("/ckpt/audioldm2-music")
    pipe.text_frontend = front
    pipe(prompt=["techno"], measurement=...)                # instead of prompt_embeds= / generated_prompt_embeds=
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


def add_special_tokens(hidden_states, attention_mask, sos_token, eos_token):
    """diffusers modeling_audioldm2.add_special_tokens: learned SOS / EOS vectors around the sequence, mask extended by ones."""
    B = hidden_states.shape[0]
    if attention_mask is not None:
        one = attention_mask.new_ones((B, 1))
        attention_mask = torch.cat([one, attention_mask, one], dim=-1)
    sos = sos_token.expand(B, 1, -1)
    eos = eos_token.expand(B, 1, -1)
    return torch.cat([sos, hidden_states, eos], dim=1), attention_mask


class AudioLDM2ProjectionModel(nn.Module):
    """diffusers 0.31.0 `AudioLDM2ProjectionModel` (pipelines/audioldm2/modeling_audioldm2.py) without the VITS position
    embedding (music checkpoints do not use it): both encoder outputs are projected to the language-model width, wrapped in
    their own learned SOS / EOS embeddings and concatenated along the sequence (reference call site :511-518)."""

    def __init__(self, text_encoder_dim, text_encoder_1_dim, langauge_model_dim):      # upstream spelling of the argument
        super().__init__()
        self.projection = nn.Linear(text_encoder_dim, langauge_model_dim)
        self.projection_1 = nn.Linear(text_encoder_1_dim, langauge_model_dim)
        self.sos_embed = nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed = nn.Parameter(torch.ones(langauge_model_dim))
        self.sos_embed_1 = nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed_1 = nn.Parameter(torch.ones(langauge_model_dim))

    def forward(self, hidden_states=None, hidden_states_1=None, attention_mask=None, attention_mask_1=None):
        hidden_states = self.projection(hidden_states)
        hidden_states, attention_mask = add_special_tokens(hidden_states, attention_mask, self.sos_embed, self.eos_embed)
        hidden_states_1 = self.projection_1(hidden_states_1)
        hidden_states_1, attention_mask_1 = add_special_tokens(hidden_states_1, attention_mask_1, self.sos_embed_1, self.eos_embed_1)
        hidden_states = torch.cat([hidden_states, hidden_states_1], dim=1)
        if attention_mask is None and attention_mask_1 is not None:
            attention_mask = attention_mask_1.new_ones((hidden_states.shape[0], hidden_states.shape[1] - attention_mask_1.shape[1]))
        elif attention_mask is not None and attention_mask_1 is None:
            attention_mask_1 = attention_mask.new_ones((hidden_states.shape[0], hidden_states.shape[1] - attention_mask.shape[1]))
        if attention_mask is not None and attention_mask_1 is not None:
            attention_mask = torch.cat([attention_mask, attention_mask_1], dim=-1)
        return hidden_states, attention_mask


def resample_to(wav, orig_sr, new_sr):
    """(B, L) fp32 -> (B, ceil(L * new / orig)) with the sinc-hann polyphase kernel the operators use (the reference calls
    librosa.resample here, :470-473; librosa is absent).  HIP FIR on the GPU, a strided conv on the host."""
    from ..inverse_problem import dsp
    if int(orig_sr) == int(new_sr):
        return wav
    kern, width, orig, new = dsp.sinc_resample_kernel(orig_sr, new_sr)
    n_out = int(np.ceil(new * wav.shape[1] / orig))
    k = torch.from_numpy(np.ascontiguousarray(kern))
    if wav.is_cuda:
        from ..inverse_problem.operator import _fir_fwd
        return _fir_fwd(wav.float().contiguous(), wav.shape[1], k.to(wav.device), n_out, orig, new, width)
    x = torch.nn.functional.pad(wav.float()[:, None], (width, width + orig))
    y = torch.nn.functional.conv1d(x, k[:, None, :], stride=orig)                  # (B, new, n)
    return y.transpose(1, 2).reshape(wav.shape[0], -1)[:, :n_out]


class AudioLDM2PromptFrontEnd:
    def __init__(self, text_encoder, tokenizer, text_encoder_2, tokenizer_2, projection_model, language_model,
                 feature_extractor=None, sampling_rate=16000):
        """text_encoder: transformers.ClapModel (get_text_features / get_audio_features); text_encoder_2: T5EncoderModel;
        language_model: GPT2Model; tokenizers callable like transformers tokenizers; feature_extractor: ClapFeatureExtractor
        (only for prompt_type="clap")."""
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.projection_model, self.language_model = projection_model, language_model
        self.feature_extractor, self.sampling_rate = feature_extractor, sampling_rate

    @classmethod
    def from_pretrained(cls, repo_dir, device="cuda", torch_dtype=torch.float32):
        from transformers import AutoTokenizer, ClapFeatureExtractor, ClapModel, GPT2Model, T5EncoderModel
        from safetensors.torch import load_file
        sub = lambda n: os.path.join(repo_dir, n)                                          # noqa: E731
        clap = ClapModel.from_pretrained(sub("text_encoder"), torch_dtype=torch_dtype).to(device).eval()
        t5 = T5EncoderModel.from_pretrained(sub("text_encoder_2"), torch_dtype=torch_dtype).to(device).eval()
        lm = GPT2Model.from_pretrained(sub("language_model"), torch_dtype=torch_dtype).to(device).eval()
        proj = AudioLDM2ProjectionModel(clap.config.projection_dim, t5.config.d_model, lm.config.n_embd)
        files = [f for f in os.listdir(sub("projection_model")) if f.endswith(".safetensors")]
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(sub("projection_model"), f)))
        proj.load_state_dict(sd, strict=True)
        fe = ClapFeatureExtractor.from_pretrained(sub("feature_extractor")) if os.path.isdir(sub("feature_extractor")) else None
        return cls(clap, AutoTokenizer.from_pretrained(sub("tokenizer")), t5, AutoTokenizer.from_pretrained(sub("tokenizer_2")),
                   proj.to(device=device, dtype=torch_dtype).eval(), lm, fe)

    def _device(self):
        try:
            return next(self.language_model.parameters()).device
        except (AttributeError, StopIteration, TypeError):
            return torch.device("cpu")

    @torch.no_grad()
    def generate_language_model(self, inputs_embeds, attention_mask=None, max_new_tokens=8):
        """plpeline_audioldm2.py:280-320: autoregressive generation in EMBEDDING space -- the last hidden state of each pass is
        appended to the input sequence; returns the `max_new_tokens` generated states.  (The reference uses the KV cache of
        `transformers`' generation utilities; re-running the prefix gives the same states and keeps this wrapper independent
        of those private helpers.)"""
        if max_new_tokens is None:
            max_new_tokens = getattr(self.language_model.config, "max_new_tokens", 8)
        for _ in range(max_new_tokens):
            out = self.language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, return_dict=True)
            inputs_embeds = torch.cat([inputs_embeds, out.last_hidden_state[:, -1:, :]], dim=1)
            if attention_mask is not None:
                attention_mask = torch.cat([attention_mask, attention_mask.new_ones((attention_mask.shape[0], 1))], dim=-1)
        return inputs_embeds[:, -max_new_tokens:, :]

    def _tokenize(self, tokenizer, texts, is_clap, max_length=None):
        return tokenizer(texts, padding="max_length" if (is_clap or max_length is not None) else True,
                         max_length=max_length if max_length is not None else tokenizer.model_max_length, truncation=True,
                         return_tensors="pt")

    @torch.no_grad()
    def _encode_one_side(self, texts, prompt_type, measurement, max_new_tokens, t5_max_length=None):
        dev = self._device()
        B = len(texts)
        # ---- encoder 1: CLAP text tower (or the CLAP audio tower on the measurement, prompt_type == "clap", :469-481)
        tin = self._tokenize(self.tokenizer, texts, True)
        ids, mask = tin.input_ids, tin.attention_mask
        untr = self.tokenizer(texts, padding="longest", return_tensors="pt").input_ids
        if untr.shape[-1] >= ids.shape[-1] and not torch.equal(ids, untr):
            removed = self.tokenizer.batch_decode(untr[:, self.tokenizer.model_max_length - 1:-1])
            logger.warning(f"The following part of your input was truncated because clap can only handle sequences up to "
                           f"{self.tokenizer.model_max_length} tokens: {removed}")
        if prompt_type == "clap":
            if measurement is None or self.feature_extractor is None:
                raise ValueError("prompt_type='clap' needs the measurement waveform and a ClapFeatureExtractor")
            target = int(self.feature_extractor.sampling_rate)
            wav = resample_to(measurement.reshape(measurement.shape[0], -1).float(), self.sampling_rate, target).cpu().numpy()
            feats = self.feature_extractor(list(wav), return_tensors="pt", sampling_rate=target).input_features
            p_dtype = next(self.text_encoder.parameters()).dtype
            e1 = self.text_encoder.get_audio_features(feats.to(device=dev, dtype=p_dtype))
        else:
            e1 = self.text_encoder.get_text_features(ids.to(dev), attention_mask=mask.to(dev))
        e1 = getattr(e1, "pooler_output", e1)
        e1 = e1[:, None, :]                                     # (B, 1, D): one hidden state to attend
        m1 = mask.new_ones((B, 1)).to(dev)
        # ---- encoder 2: T5
        tin2 = self._tokenize(self.tokenizer_2, texts, False, t5_max_length)
        ids2, m2 = tin2.input_ids.to(dev), tin2.attention_mask.to(dev)
        e2 = self.text_encoder_2(ids2, attention_mask=m2)[0]
        # ---- projection + GPT-2 generation
        h, hm = self.projection_model(hidden_states=e1, hidden_states_1=e2, attention_mask=m1, attention_mask_1=m2)
        gen = self.generate_language_model(h, attention_mask=hm, max_new_tokens=max_new_tokens)
        return e2.float(), m2, gen.float()

    def encode(self, prompt, negative_prompt=None, do_classifier_free_guidance=True, max_new_tokens=None, prompt_type=None,
               measurement=None, transcription=None):
        """-> dict(prompt_embeds, attention_mask, generated_prompt_embeds[, negative_*]); repetition per waveform and the
        [uncond | text] concatenation stay in the pipeline (`_prepare_cond`)."""
        if isinstance(prompt, str):
            prompt = [prompt]
        elif not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        B = len(prompt)
        pe, am, ge = self._encode_one_side(prompt, prompt_type, measurement, max_new_tokens)
        out = dict(prompt_embeds=pe, attention_mask=am, generated_prompt_embeds=ge)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                uncond = [""] * B
            elif isinstance(negative_prompt, str):
                if B != 1:
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {list}.")
                uncond = [negative_prompt]
            elif B != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} has"
                                 f" batch size {B}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
            else:
                uncond = list(negative_prompt)
            # the unconditional T5 sequence is padded to the conditional one's length (:572-580); its CLAP branch is always text
            npe, nam, nge = self._encode_one_side(uncond, None, None, max_new_tokens, t5_max_length=pe.shape[1])
            out.update(negative_prompt_embeds=npe, negative_attention_mask=nam, negative_generated_prompt_embeds=nge)
        return out
