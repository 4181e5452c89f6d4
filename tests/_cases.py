"""Case tables shared by tests/golden/gen_golden.py and the tests (pure data)."""

LLL, DDD = ("l", "l", "l"), ("d", "d", "d")

# (shape THW, cuboid, strategy)
REORDER_CASES = [
    ((13, 16, 16), (13, 1, 1), LLL),
    ((13, 16, 16), (1, 16, 1), LLL),
    ((13, 16, 16), (1, 1, 16), LLL),
    ((14, 16, 16), (2, 4, 4), LLL),
    ((14, 16, 16), (2, 4, 4), DDD),
    ((14, 16, 16), (2, 4, 4), ("l", "d", "d")),
    ((6, 8, 12), (3, 2, 4), ("d", "l", "d")),
]

# (shape THW, cuboid, shift, strategy, padding_type)
MASK_CASES = [
    ((13, 16, 16), (13, 1, 1), (0, 0, 0), LLL, "zeros"),
    ((13, 16, 16), (2, 4, 4), (0, 0, 0), LLL, "zeros"),
    ((13, 16, 16), (2, 4, 4), (1, 2, 2), LLL, "zeros"),
    ((13, 16, 16), (2, 4, 4), (1, 2, 2), LLL, "ignore"),
    ((13, 16, 16), (2, 4, 4), (0, 0, 0), LLL, "ignore"),
    ((5, 7, 6), (2, 4, 4), (1, 2, 2), LLL, "ignore"),
    ((5, 7, 6), (2, 4, 4), (1, 2, 2), LLL, "nearest"),
    ((5, 8, 8), (1, 4, 4), (0, 0, 0), DDD, "zeros"),
    ((5, 8, 8), (1, 4, 4), (0, 2, 2), DDD, "ignore"),       # shift is dropped for 'd'
    ((3, 8, 8), (4, 16, 2), (2, 1, 1), LLL, "ignore"),      # cuboid clamped to the data size
]


def _attn(shape, cuboid, shift=(0, 0, 0), strategy=LLL, padding_type="zeros", dim=32, heads=2, B=2):
    return dict(shape=shape, cuboid=cuboid, shift=shift, strategy=strategy, padding_type=padding_type,
                dim=dim, heads=heads, B=B)


ATTN_CASES = [
    _attn((5, 8, 8), (5, 1, 1)),                                        # axial T
    _attn((5, 8, 8), (1, 8, 1)),                                        # axial H
    _attn((5, 8, 8), (1, 1, 8)),                                        # axial W
    _attn((5, 8, 8), (2, 4, 4)),                                        # video_swin, pad_t = 1 ('zeros': pad keeps mass)
    _attn((5, 8, 8), (2, 4, 4), shift=(1, 2, 2)),                       # shifted window
    _attn((5, 8, 8), (2, 4, 4), shift=(1, 2, 2), padding_type="ignore"),
    _attn((5, 7, 6), (2, 4, 4), shift=(1, 2, 2), padding_type="ignore"),
    _attn((5, 7, 6), (2, 4, 4), padding_type="nearest"),
    _attn((5, 8, 8), (1, 4, 4), strategy=DDD),                          # dilated
    _attn((3, 8, 8), (4, 16, 2), shift=(2, 1, 1), padding_type="ignore"),   # clamped cuboid, rel-pos slice quirk Q2
    _attn((13, 16, 16), (13, 1, 1), dim=256, heads=4, B=1),             # v1 level-0 shapes
    _attn((13, 16, 16), (1, 16, 1), dim=256, heads=4, B=1),
    _attn((13, 8, 8), (1, 1, 8), dim=512, heads=4, B=1),                # v1 level-1 shape (head_dim 128)
]

# TimeEmbedResBlock cases: channels-last shapes (T, H, W)
RESBLOCK3D_CASES = [
    dict(cin=32, cout=32, emb=64, ssn=False, shape=(3, 6, 6)),
    dict(cin=32, cout=32, emb=64, ssn=True, shape=(3, 6, 6)),
    dict(cin=5, cout=32, emb=None, ssn=False, shape=(5, 8, 8)),        # first_proj-like: 5 groups of 1, 1x1x1 skip
    dict(cin=32, cout=64, emb=64, ssn=False, shape=(3, 4, 4)),
]


def _unet(pattern=None, **over):
    cfg = dict(input_shape=[3, 8, 8, 4], target_shape=[2, 8, 8, 4], base_units=64, scale_alpha=1.0,
               depth=[1, 1], downsample=2, downsample_type="patch_merge", upsample_type="upsample",
               upsample_kernel_size=3, block_attn_patterns=pattern, num_heads=2,
               attn_drop=0.0, proj_drop=0.0, ffn_drop=0.0, ffn_activation="gelu", gated_ffn=False,
               norm_layer="layer_norm", use_inter_ffn=True, hierarchical_pos_embed=False,
               pos_embed_type="t+h+w", padding_type="zeros", checkpoint_level=0, use_relative_pos=True,
               self_attn_use_final_proj=True, num_global_vectors=0, time_embed_channels_mult=4,
               time_embed_use_scale_shift_norm=False, time_embed_dropout=0.0, unet_res_connect=True)
    cfg.update(over)
    return cfg


TINY_UNET_CFGS = {
    "axial": _unet("axial"),
    "video_swin_2x4": _unet("video_swin_2x4"),
    "spatial_lg_4": _unet("spatial_lg_4", padding_type="ignore"),
    "axial_depth2": _unet("axial", depth=[2, 2], ffn_activation="leaky"),
    "default_cuboids": _unet(None, padding_type="ignore", use_inter_ffn=False),   # ctor-default (4,4,4) l / d
    # cuboids larger than 64 slots: one cuboid = the whole grid (5 x 8 x 8 = 320, level 1: 80) / one frame (16 x 16 = 256, level 1: 64)
    "full": _unet("full"),
    "divided_st_16": _unet("divided_st", input_shape=[3, 16, 16, 4], target_shape=[2, 16, 16, 4]),
}

TINY_VAE_CFG = dict(in_channels=1, out_channels=1, down_block_types=["DownEncoderBlock2D"] * 3,
                    up_block_types=["UpDecoderBlock2D"] * 3, block_out_channels=[32, 64, 64],
                    layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=8)

# v1 / N-body stand-in / full-res constructor keyword sets live in the package (bench.py uses them too)
from prediff_amd.presets import (FULLRES_UNET_CFG, NBODY_LDM_KW, NBODY_UNET_CFG, NBODY_VAE_CFG, V1_LDM_KW,  # noqa: E402,F401
                                 V1_UNET_CFG, V1_VAE_CFG)


def _align(**over):
    cfg = dict(input_shape=[2, 8, 8, 4], out_channels=1, base_units=32, scale_alpha=1.0, depth=[1, 1], downsample=2,
               downsample_type="patch_merge", block_attn_patterns="axial", num_heads=2, attn_drop=0.0, proj_drop=0.0,
               ffn_drop=0.0, ffn_activation="gelu", gated_ffn=False, norm_layer="layer_norm", use_inter_ffn=True,
               hierarchical_pos_embed=False, pos_embed_type="t+h+w", padding_type="zeros", checkpoint_level=0,
               use_relative_pos=True, self_attn_use_final_proj=True, num_global_vectors=0, time_embed_channels_mult=4,
               time_embed_use_scale_shift_norm=False, time_embed_dropout=0.0, pool="attention", readout_seq=True, out_len=2)
    cfg.update(over)
    return cfg


TINY_ALIGN_ARGS = _align()
# prediff_sevirlr_v1.yaml:104-155 (model.align.model_args)
V1_ALIGN_ARGS = _align(input_shape=[6, 16, 16, 64], base_units=128, num_heads=4, attn_drop=0.1, proj_drop=0.1, ffn_drop=0.1,
                       out_len=6)
