"""Prompt-conditioning front end for MusicLDM: wraps -- does not re-implement -- a
`transformers` CLAP text tower and its tokenizer, following `MusicLDMPipeline._encode_prompt`
(diffmusic/pipelines/pipeline_musicldm.py:119-250).  Runs once per call, outside the hot loop; whatever device the wrapped
encoder lives on is used for it.  The engine itself only ever sees the resulting (B, 512) embeddings.

    front = ClapTextFrontEnd.from_pretrained("/ckpt/musicldm")        # text_encoder/ + tokenizer/ sub-folders
    pipe.text_frontend = front
    pipe(prompt=["a piano"], measurement=...)                          # instead of prompt_embeds=...
"""
import logging
import os

import torch

logger = logging.getLogger(__name__)


class ClapTextFrontEnd:
    def __init__(self, text_encoder, tokenizer):
        """text_encoder: object with get_text_features(input_ids, attention_mask=...) -> (B, D) (transformers.ClapModel);
        tokenizer: callable like a transformers tokenizer (padding / max_length / truncation / return_tensors) with
        `model_max_length` and `batch_decode`."""
        self.text_encoder, self.tokenizer = text_encoder, tokenizer

    @classmethod
    def from_pretrained(cls, repo_dir, device="cuda", torch_dtype=torch.float32):
        from transformers import AutoTokenizer, ClapModel
        enc = ClapModel.from_pretrained(os.path.join(repo_dir, "text_encoder"), torch_dtype=torch_dtype).to(device).eval()
        tok = AutoTokenizer.from_pretrained(os.path.join(repo_dir, "tokenizer"))
        return cls(enc, tok)

    def _device(self):
        try:
            return next(self.text_encoder.parameters()).device
        except (AttributeError, StopIteration, TypeError):
            return torch.device("cpu")

    @torch.no_grad()
    def _embed(self, texts, max_length):
        dev = self._device()
        inputs = self.tokenizer(texts, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt")
        ids, mask = inputs.input_ids, inputs.attention_mask
        untruncated = self.tokenizer(texts, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ids.shape[-1] and not torch.equal(ids, untruncated):
            removed = self.tokenizer.batch_decode(untruncated[:, max_length - 1:-1])
            logger.warning("The following part of your input was truncated because CLAP can only handle sequences up to"
                           f" {max_length} tokens: {removed}")
        return self.text_encoder.get_text_features(ids.to(dev), attention_mask=mask.to(dev)).float()

    def encode(self, prompt, negative_prompt=None, do_classifier_free_guidance=True):
        """-> (prompt_embeds (B, D), negative_prompt_embeds (B, D) or None); repetition per waveform and the [uncond | text]
        concatenation stay in the pipeline (`_prepare_cond`)."""
        if isinstance(prompt, str):
            prompt = [prompt]
        elif not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        batch_size = len(prompt)
        pe = self._embed(prompt, self.tokenizer.model_max_length)
        ne = None
        if do_classifier_free_guidance:
            if negative_prompt is None:
                uncond = [""] * batch_size
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt] * (batch_size if batch_size == 1 else 1)
                if batch_size != 1:
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {list}.")
            elif not isinstance(negative_prompt, list):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {list}.")
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} has"
                                 f" batch size {batch_size}. Please make sure that passed `negative_prompt` matches the batch size of"
                                 " `prompt`.")
            else:
                uncond = negative_prompt
            ne = self._embed(uncond, self.tokenizer.model_max_length)
        return pe, ne
