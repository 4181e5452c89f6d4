"""Constructor keyword sets of the configurations BASELINE.json names (pure data; no compute).

V1_*: scripts/prediff/sevirlr/prediff_sevirlr_v1.yaml mapped through scripts/prediff/sevirlr/train_sevirlr_prediff.py:91-137
(the same mapping `prediff_amd.config.build_prediff` applies to the YAML; dropouts only matter in training).
NBODY_*: the N-body-MNIST stand-in of BASELINE config 1 (the reference ships no such config, F8).
FULLRES_*: BASELINE config 5 (SEVIR full-res 384x384, 13 -> 12 frames; no reference config either).
"""
import copy

# prediff_sevirlr_v1.yaml:157-205
V1_UNET_CFG = dict(input_shape=[7, 16, 16, 64], target_shape=[6, 16, 16, 64], base_units=256, scale_alpha=1.0,
                   depth=[4, 4], downsample=2, downsample_type="patch_merge", upsample_type="upsample",
                   upsample_kernel_size=3, block_attn_patterns="axial", num_heads=4, attn_drop=0.1,
                   proj_drop=0.1, ffn_drop=0.1, ffn_activation="gelu", gated_ffn=False, norm_layer="layer_norm",
                   use_inter_ffn=True, hierarchical_pos_embed=False, pos_embed_type="t+h+w",
                   padding_type="zeros", checkpoint_level=0, use_relative_pos=True,
                   self_attn_use_final_proj=True, num_global_vectors=0, use_global_vector_ffn=False,
                   use_global_self_attn=True, separate_global_qkv=True, global_dim_ratio=1,
                   attn_linear_init_mode="0", ffn_linear_init_mode="0", ffn2_linear_init_mode="2",
                   attn_proj_linear_init_mode="2", conv_init_mode="0", down_linear_init_mode="0",
                   up_linear_init_mode="0", global_proj_linear_init_mode="2", norm_init_mode="0",
                   time_embed_channels_mult=4, time_embed_use_scale_shift_norm=False, time_embed_dropout=0.0,
                   unet_res_connect=True)

# prediff_sevirlr_v1.yaml:206-217
V1_VAE_CFG = dict(in_channels=1, out_channels=1, down_block_types=["DownEncoderBlock2D"] * 4,
                  up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512],
                  layers_per_block=2, act_fn="silu", latent_channels=64, norm_num_groups=32)

# prediff_sevirlr_v1.yaml:104-155 (model.align.model_args)
V1_ALIGN_ARGS = dict(input_shape=[6, 16, 16, 64], out_channels=1, base_units=128, scale_alpha=1.0, depth=[1, 1], downsample=2,
                     downsample_type="patch_merge", block_attn_patterns="axial", num_heads=4, attn_drop=0.1, proj_drop=0.1,
                     ffn_drop=0.1, ffn_activation="gelu", gated_ffn=False, norm_layer="layer_norm", use_inter_ffn=True,
                     hierarchical_pos_embed=False, pos_embed_type="t+h+w", padding_type="zeros", checkpoint_level=0,
                     use_relative_pos=True, self_attn_use_final_proj=True, num_global_vectors=0, time_embed_channels_mult=4,
                     time_embed_use_scale_shift_norm=False, time_embed_dropout=0.0, pool="attention", readout_seq=True, out_len=6)

V1_LDM_KW = dict(layout="NTHWC", data_shape=(6, 128, 128, 1), timesteps=1000, beta_schedule="linear", use_ema=False,
                 latent_shape=(6, 16, 16, 64), scale_factor=1.0)

# BASELINE config 1 stand-in: 10 context + 10 future frames of 64x64, VAE /4 -> 16x16x4 latents, small axial denoiser, DDIM-10
NBODY_VAE_CFG = dict(in_channels=1, out_channels=1, down_block_types=["DownEncoderBlock2D"] * 3,
                     up_block_types=["UpDecoderBlock2D"] * 3, block_out_channels=[32, 64, 64],
                     layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=8)
NBODY_UNET_CFG = copy.deepcopy(V1_UNET_CFG)
NBODY_UNET_CFG.update(input_shape=[10, 16, 16, 4], target_shape=[10, 16, 16, 4], base_units=32, depth=[1, 1], num_heads=2,
                      attn_drop=0.0, proj_drop=0.0, ffn_drop=0.0)
NBODY_LDM_KW = dict(layout="NTHWC", data_shape=(10, 64, 64, 1), timesteps=1000, beta_schedule="linear", use_ema=False,
                    latent_shape=(10, 16, 16, 4), scale_factor=1.0)

# BASELINE config 5: 384x384 frames, 13 context -> 12 future, VAE /8 -> latent (13|12, 48, 48, 64); the v1 denoiser on the larger
# grid (axial cuboids 25 / 48 / 48 at level 0, 25 / 24 / 24 at level 1)
FULLRES_UNET_CFG = copy.deepcopy(V1_UNET_CFG)
FULLRES_UNET_CFG.update(input_shape=[13, 48, 48, 64], target_shape=[12, 48, 48, 64])
FULLRES_LDM_KW = dict(layout="NTHWC", data_shape=(12, 384, 384, 1), timesteps=1000, beta_schedule="linear", use_ema=False,
                      latent_shape=(12, 48, 48, 64), scale_factor=1.0)
