"""CPU: YAML front end: scalar coercion, kwargs mapping, schema of the built modules."""
import json
import os

import pytest
import torch

from _cases import V1_ALIGN_ARGS, V1_UNET_CFG, V1_VAE_CFG
from prediff_amd import config as CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_yaml_exponent_scalars_are_coerced():
    import yaml
    raw = yaml.safe_load(open(os.path.join(ROOT, "configs", "prediff_sevirlr_v1.yaml")))
    assert isinstance(raw["model"]["diffusion"]["linear_start"], str)          # the PyYAML trap ...
    cfg = CFG.load_config(os.path.join(ROOT, "configs", "prediff_sevirlr_v1.yaml"))
    d = cfg["model"]["diffusion"]
    assert d["linear_start"] == 1e-4 and d["linear_end"] == 2e-2 and d["cosine_s"] == 8e-3   # ... is handled
    assert CFG._coerce("axial") == "axial" and CFG._coerce("1e-3") == 1e-3 and CFG._coerce("-2.5E+2") == -250.0


def test_kwargs_mapping_matches_the_script():
    cfg = CFG.load_config(os.path.join(ROOT, "configs", "prediff_sevirlr_v1.yaml"))
    kw = CFG.unet_kwargs(cfg["model"]["latent_model"])
    ref = dict(V1_UNET_CFG)
    ref["block_attn_patterns"] = ["axial", "axial"]
    for k in ("hierarchical_pos_embed", "use_inter_ffn"):      # never passed by the script: constructor defaults apply
        assert ref.pop(k) == {"hierarchical_pos_embed": False, "use_inter_ffn": True}[k]
    assert kw == ref
    assert CFG.vae_kwargs(cfg["model"]["vae"]) == V1_VAE_CFG
    dk = CFG.diffusion_kwargs(cfg)
    assert dk["timesteps"] == 1000 and dk["layout"] == "NTHWC" and dk["latent_shape"] == (6, 16, 16, 64) and "latent_cond_shape" not in dk
    assert cfg["model"]["align"]["model_args"] == V1_ALIGN_ARGS


def test_build_from_config_on_cpu_gives_reference_schema():
    cfg = CFG.load_config(os.path.join(ROOT, "configs", "prediff_sevirlr_v1.yaml"))
    ldm, align = CFG.build_prediff(cfg, device="cpu")
    ref_u = json.load(open(os.path.join(GOLDEN, "v1_unet_schema.json")))
    ref_v = json.load(open(os.path.join(GOLDEN, "v1_vae_schema.json")))
    ref_a = json.load(open(os.path.join(GOLDEN, "v1_align_schema.json")))
    assert list(ldm.torch_nn_module.state_dict().keys()) == list(ref_u.keys())
    assert list(ldm.first_stage_model.state_dict().keys()) == list(ref_v.keys())
    assert list(align.model.state_dict().keys()) == list(ref_a.keys())
    assert ldm.num_timesteps == 1000 and ldm.alignment_fn is not None and align.guide_scale == 50.0
    assert abs(float(ldm.betas[0]) - 1e-4) < 1e-10 and abs(float(ldm.betas[-1]) - 2e-2) < 1e-8
    ctx, tgt = CFG.split_sequence(torch.zeros(2, 13, 4, 4, 1), 7, 6)
    assert ctx.shape[1] == 7 and tgt.shape[1] == 6
