"""Named cuboid self-attention patterns: the ``self_pattern`` / ``block_attn_patterns`` config surface.

Same names and results as the reference registry (models/cuboid_transformer/cuboid_transformer_patterns.py:
full :11-16, axial :19-37, video_swin[_PxM] :40-50,66-72, divided_st :53-58, spatial_lg_v1 / spatial_lg_M :76-97,
axial_space_dilate_K :100-118).  ``CuboidSelfAttentionPatterns.get(name)(mem_shape)`` returns
(cuboid_sizes, strategies, shift_sizes) like the reference's ``Registry.get``.
"""
from typing import Callable, Dict

_LLL, _DDD, _Z = ("l", "l", "l"), ("d", "d", "d"), (0, 0, 0)


class _PatternRegistry:
    def __init__(self, name):
        self.name = name
        self._fns: Dict[str, Callable] = {}

    def register(self, key, fn=None):
        if fn is None:
            def deco(f):
                self._fns[key] = f
                return f
            return deco
        self._fns[key] = fn
        return fn

    def get(self, key):
        if key not in self._fns:
            raise KeyError(f"{key!r} is not a registered {self.name}; known: {sorted(self._fns)}")
        return self._fns[key]

    def list_keys(self):
        return list(self._fns)


CuboidSelfAttentionPatterns = _PatternRegistry("CuboidSelfAttentionPattern")


def full_attention(shape):
    T, H, W = shape[:3]
    return [(T, H, W)], [_LLL], [_Z]


def self_axial(shape):
    T, H, W = shape[:3]
    return [(T, 1, 1), (1, H, 1), (1, 1, W)], [_LLL] * 3, [_Z] * 3


def self_video_swin(shape, P=2, M=4):
    T, H, W = shape[:3]
    P, M = min(P, T), min(M, H, W)
    return [(P, M, M), (P, M, M)], [_LLL] * 2, [_Z, (P // 2, M // 2, M // 2)]


def self_divided_space_time(shape):
    T, H, W = shape[:3]
    return [(T, 1, 1), (1, H, W)], [_LLL] * 2, [_Z] * 2


def self_spatial_lg_v1(shape, M=4):
    T, H, W = shape[:3]
    if H <= M and W <= M:
        return [(T, 1, 1), (1, H, W)], [_LLL] * 2, [_Z] * 2
    return [(T, 1, 1), (1, M, M), (1, M, M)], [_LLL, _LLL, _DDD], [_Z] * 3


def self_axial_space_dilate_K(shape, K=2):
    T, H, W = shape[:3]
    K = min(K, H, W)
    return ([(T, 1, 1), (1, H // K, 1), (1, H // K, 1), (1, 1, W // K), (1, 1, W // K)],
            [_LLL, _DDD, _LLL, _DDD, _LLL], [_Z] * 5)


def _bind(fn, **kw):
    return lambda shape: fn(shape, **kw)


CuboidSelfAttentionPatterns.register("full", full_attention)
CuboidSelfAttentionPatterns.register("axial", self_axial)
CuboidSelfAttentionPatterns.register("video_swin", self_video_swin)
CuboidSelfAttentionPatterns.register("divided_st", self_divided_space_time)
for _p in (1, 2, 4, 8, 10):
    for _m in (1, 2, 4, 8, 16, 32):
        CuboidSelfAttentionPatterns.register(f"video_swin_{_p}x{_m}", _bind(self_video_swin, P=_p, M=_m))
CuboidSelfAttentionPatterns.register("spatial_lg_v1", self_spatial_lg_v1)
for _m in (1, 2, 4, 8, 16, 32):
    CuboidSelfAttentionPatterns.register(f"spatial_lg_{_m}", _bind(self_spatial_lg_v1, M=_m))
for _k in (2, 4, 8):
    CuboidSelfAttentionPatterns.register(f"axial_space_dilate_{_k}", _bind(self_axial_space_dilate_K, K=_k))
