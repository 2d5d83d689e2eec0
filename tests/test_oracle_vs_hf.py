"""Independent second opinion for the oracle: the oracle's
towers vs ``transformers.CLIPModel`` (an unrelated implementation of OpenAI CLIP)
built from a local config with the SAME random weights.  CPU only, no downloads."""
import pytest
import torch

from clipfs import synth
from oracle import clip_oracle as O

transformers = pytest.importorskip("transformers")


def _to_hf(sd, cfg):
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig
    tc = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.transformer_width,
                        intermediate_size=4 * cfg.transformer_width, num_hidden_layers=cfg.transformer_layers,
                        num_attention_heads=cfg.transformer_heads, max_position_embeddings=cfg.context_length,
                        hidden_act="quick_gelu", projection_dim=cfg.embed_dim,
                        eos_token_id=cfg.vocab_size - 1, bos_token_id=cfg.vocab_size - 2, pad_token_id=0)
    vc = CLIPVisionConfig(hidden_size=cfg.vision_width, intermediate_size=4 * cfg.vision_width,
                          num_hidden_layers=cfg.vision_layers, num_attention_heads=cfg.vision_heads,
                          image_size=cfg.image_resolution, patch_size=cfg.vision_patch_size,
                          hidden_act="quick_gelu", projection_dim=cfg.embed_dim)
    hf = CLIPModel(CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(),
                              projection_dim=cfg.embed_dim)).double().eval()
    new = {}
    new["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    new["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    new["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    new["vision_model.pre_layrnorm.weight"] = sd["visual.ln_pre.weight"]
    new["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.bias"]
    new["vision_model.post_layernorm.weight"] = sd["visual.ln_post.weight"]
    new["vision_model.post_layernorm.bias"] = sd["visual.ln_post.bias"]
    new["visual_projection.weight"] = sd["visual.proj"].t()
    new["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    new["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    new["text_model.final_layer_norm.weight"] = sd["ln_final.weight"]
    new["text_model.final_layer_norm.bias"] = sd["ln_final.bias"]
    new["text_projection.weight"] = sd["text_projection"].t()
    new["logit_scale"] = sd["logit_scale"]
    for src, dst, n, w in (("visual.transformer", "vision_model", cfg.vision_layers, cfg.vision_width),
                           ("transformer", "text_model", cfg.transformer_layers, cfg.transformer_width)):
        for i in range(n):
            s, d = f"{src}.resblocks.{i}.", f"{dst}.encoder.layers.{i}."
            wi, bi = sd[s + "attn.in_proj_weight"], sd[s + "attn.in_proj_bias"]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                new[d + f"self_attn.{nm}.weight"] = wi[j * w:(j + 1) * w]
                new[d + f"self_attn.{nm}.bias"] = bi[j * w:(j + 1) * w]
            new[d + "self_attn.out_proj.weight"] = sd[s + "attn.out_proj.weight"]
            new[d + "self_attn.out_proj.bias"] = sd[s + "attn.out_proj.bias"]
            for a, b in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"),
                         ("mlp.c_proj", "mlp.fc2")):
                new[d + b + ".weight"] = sd[s + a + ".weight"]
                new[d + b + ".bias"] = sd[s + a + ".bias"]
    missing, unexpected = hf.load_state_dict({k: v.double() for k, v in new.items()}, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    return hf


@pytest.mark.parametrize("cfg", [synth.TINY, synth.SMALL], ids=lambda c: c.name)
def test_oracle_matches_hf_clip(cfg):
    sd = {k: v.double() for k, v in synth.synth_state_dict(cfg, seed=7, perturb=True).items()}
    hf = _to_hf(sd, cfg)
    img = synth.synth_images(3, cfg.image_resolution, seed=3).double()
    txt = synth.synth_captions(5, cfg.context_length, cfg.vocab_size, seed=4, max_len=10)
    with torch.no_grad():
        fi = O.encode_image(sd, img)
        ft = O.encode_text(sd, txt)
        hi = hf.get_image_features(pixel_values=img)
        ht = hf.get_text_features(input_ids=txt, attention_mask=torch.ones_like(txt))
        hi = getattr(hi, "pooler_output", hi)
        ht = getattr(ht, "pooler_output", ht)
        li, _ = O.clip_forward(sd, img, txt)
        out = hf(input_ids=txt, pixel_values=img, attention_mask=torch.ones_like(txt))
    assert torch.allclose(fi, hi, atol=1e-10, rtol=1e-9), (fi - hi).abs().max()
    assert torch.allclose(ft, ht, atol=1e-10, rtol=1e-9), (ft - ht).abs().max()
    assert torch.allclose(li, out.logits_per_image, atol=1e-8, rtol=1e-9)
