"""-m gpu: SURVEY.md section 8f row 2 on the GPU box -- the AudioLDM2 prompt front end (diffmusic_amd/pipelines/prompt_audioldm2.py)
on `cuda` feeding the REAL HIP AudioLDM2 U-Net (production widths: GPT-2 states 768, T5 states 1024, CLAP projection 512):

  * `AudioLDM2Pipeline(prompt=...)` through `AudioLDM2PromptFrontEnd` equals passing the front end's tensors by hand
    (reference: diffmusic/pipelines/plpeline_audioldm2.py:1077-1102, 1147-1154),
  * the embedding-space GPT-2 generation loop equals a plain `transformers` forward with `past_key_values` at 8 new tokens
    (plpeline_audioldm2.py:280-320),
  * `prompt_type="clap"`: the measurement waveform on the GPU goes through the HIP resampler and the CLAP audio tower (:469-481).

The encoders are randomly initialised `transformers` modules of the production WIDTHS with few layers (no checkpoints offline)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_prompt_audioldm2 import Tok                                     # noqa: E402  (tokenizer stand-in: no vocabulary files offline)


@pytest.fixture(scope="module")
def front():
    from transformers import (ClapAudioConfig, ClapConfig, ClapFeatureExtractor, ClapModel, ClapTextConfig, GPT2Config, GPT2Model,
                              T5Config, T5EncoderModel)
    from diffmusic_amd.pipelines.prompt_audioldm2 import AudioLDM2ProjectionModel, AudioLDM2PromptFrontEnd
    torch.manual_seed(0)
    clap = ClapModel(ClapConfig(text_config=ClapTextConfig(vocab_size=60, hidden_size=768, num_hidden_layers=2, num_attention_heads=12,
                                                           intermediate_size=3072, max_position_embeddings=80, projection_dim=512).to_dict(),
                                audio_config=ClapAudioConfig(depths=[1, 1, 2, 1], projection_dim=512).to_dict(), projection_dim=512))
    t5 = T5EncoderModel(T5Config(vocab_size=60, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16,
                                 feed_forward_proj="gated-gelu"))
    lm = GPT2Model(GPT2Config(vocab_size=8, n_positions=128, n_embd=768, n_layer=2, n_head=12))
    proj = AudioLDM2ProjectionModel(512, 1024, 768)
    with torch.no_grad():                                  # learned SOS / EOS vectors: anything but the all-ones initial value
        for p in (proj.sos_embed, proj.eos_embed, proj.sos_embed_1, proj.eos_embed_1):
            p.copy_(0.05 * torch.randn(p.shape))
    fe = ClapFeatureExtractor(truncation="rand_trunc")
    mods = [m.to("cuda").eval() for m in (clap, t5, proj, lm)]
    return AudioLDM2PromptFrontEnd(mods[0], Tok(64), mods[1], Tok(24), mods[2], mods[3], feature_extractor=fe)


def test_gpt2_embedding_space_generation_equals_kv_cached_forward_on_gpu(front):
    g = torch.Generator().manual_seed(1)
    x = (0.3 * torch.randn(2, 21, 768, generator=g)).cuda()
    mask = torch.ones(2, 21, dtype=torch.long, device="cuda")
    mask[1, 15:] = 0                                        # a padded T5 tail, as the projection model hands over
    gen = front.generate_language_model(x, attention_mask=mask, max_new_tokens=8)
    assert gen.shape == (2, 8, 768) and gen.is_cuda
    with torch.no_grad():
        out = front.language_model(inputs_embeds=x, attention_mask=mask, use_cache=True, return_dict=True)
        states, past, m = [out.last_hidden_state[:, -1:]], out.past_key_values, mask
        for _ in range(7):
            m = torch.cat([m, m.new_ones(2, 1)], dim=-1)
            out = front.language_model(inputs_embeds=states[-1], attention_mask=m, past_key_values=past, use_cache=True, return_dict=True)
            states.append(out.last_hidden_state[:, -1:])
            past = out.past_key_values
    ref = torch.cat(states, dim=1)
    rel = float((gen - ref).norm() / ref.norm())
    assert rel < 1e-4, rel


def test_audioldm2_pipeline_prompt_through_front_end_on_gpu(front):
    """`pipe(prompt=...)` == passing the tensors, with the real AudioLDM2 HIP U-Net attending both contexts."""
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.schedulers import get_scheduler
    from diffmusic_amd import inverse_problem as P
    import bench
    pipe = get_pipeline("audioldm2").from_pretrained("synthetic", seed=0).to("cuda")
    op = P.PhaseRetrievalOperator(noiser=P.get_noiser("gaussian", 0.0))
    pipe.scheduler = get_scheduler("dsg")(operator=op, **bench.SCHED_CFG)
    L = 40960                                               # 2.56 s
    g = torch.Generator().manual_seed(2)
    meas = op.forward((0.2 * torch.randn(2, L, generator=g)).cuda())
    prompts = ["a slow jazz trio with brushed drums", "techno"]
    enc = front.encode(prompts, None, True, max_new_tokens=8)
    assert enc["prompt_embeds"].is_cuda and enc["prompt_embeds"].shape[0] == 2 and enc["prompt_embeds"].shape[2] == 1024
    assert enc["generated_prompt_embeds"].shape == (2, 8, 768)
    assert int(enc["attention_mask"][1].sum()) < int(enc["attention_mask"][0].sum())       # the short prompt has masked T5 keys
    assert not torch.allclose(enc["generated_prompt_embeds"], enc["negative_generated_prompt_embeds"])
    kw = dict(audio_length_in_s=2.56, num_inference_steps=3, guidance_scale=3.5, eta=1.0, ip_guidance_rate=0.08, measurement=meas,
              show_progress=False, output_type="latent")
    gens = lambda: [torch.Generator().manual_seed(k) for k in range(2)]                      # noqa: E731
    a = pipe(prompt_embeds=enc["prompt_embeds"], attention_mask=enc["attention_mask"],
             generated_prompt_embeds=enc["generated_prompt_embeds"], negative_prompt_embeds=enc["negative_prompt_embeds"],
             negative_attention_mask=enc["negative_attention_mask"],
             negative_generated_prompt_embeds=enc["negative_generated_prompt_embeds"], generator=gens(), **kw).audios
    pipe.text_frontend = front
    b = pipe(prompt=prompts, generator=gens(), **kw).audios
    assert a.shape == (2, 8, 64, 16) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b)
    assert len(pipe.last_losses) == 3 and pipe.nan_restarts == 0
    # the conditioning reaches the U-Net: another prompt moves the latents
    c = pipe(prompt=["solo violin", "techno"], generator=gens(), **kw).audios
    assert not torch.equal(c[0], b[0])


def test_clap_audio_prompt_type_on_gpu(front):
    g = torch.Generator().manual_seed(3)
    meas = (0.1 * torch.randn(2, 32000, generator=g)).cuda()
    a = front.encode(["x", "y"], None, False, prompt_type="clap", measurement=meas)
    b = front.encode(["x", "y"], None, False)
    assert a["generated_prompt_embeds"].is_cuda and a["generated_prompt_embeds"].shape == (2, 8, 768)
    assert bool(torch.isfinite(a["generated_prompt_embeds"]).all())
    assert not torch.allclose(a["generated_prompt_embeds"], b["generated_prompt_embeds"])   # audio tower instead of the text tower
    assert torch.equal(a["prompt_embeds"], b["prompt_embeds"])                              # the T5 branch is unchanged
