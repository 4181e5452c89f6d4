"""Oracle: SEVIR skill-score counts and scores (CPU, numpy).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates datasets/sevir/evaluation.py: _threshold :12-38 (>= T, NaN in either input zeroes both), update :233-239,
calc_seq_hits_misses_fas :193-211, preprocess "sevir" :213-219 (x / fp32(1/255), datasets/sevir/sevir_dataloader.py:679),
compute :241-285 (pod/sucr/csi/bias :172-191).  Integer counts are exact.
"""
import numpy as np

THRESHOLDS = (16, 74, 133, 160, 181, 219)


def counts(pred: np.ndarray, target: np.ndarray, t_axis: int, thresholds=THRESHOLDS, keep_seq=True):
    """hits/misses/false-alarms, shape (n_thresholds, T) if keep_seq else (n_thresholds,)."""
    scale = np.float32(1.0 / 255.0)
    p = pred.astype(np.float32) / scale
    t = target.astype(np.float32) / scale
    nan = np.isnan(p) | np.isnan(t)
    axes = tuple(a for a in range(pred.ndim) if not (keep_seq and a == t_axis))
    out = []
    for T in thresholds:
        tb = (t >= T) & ~nan
        pb = (p >= T) & ~nan
        out.append([np.sum(tb & pb, axis=axes), np.sum(tb & ~pb, axis=axes), np.sum(~tb & pb, axis=axes)])
    out = np.asarray(out, dtype=np.int64)            # (thr, 3, [T])
    return out[:, 0], out[:, 1], out[:, 2]


def scores(hits, misses, fas, eps=1e-4):
    hits, misses, fas = (np.asarray(v, dtype=np.float32) for v in (hits, misses, fas))
    eps = np.float32(eps)
    bias = (hits + fas) / (hits + misses + eps)
    return {"pod": hits / (hits + misses + eps), "sucr": hits / (hits + fas + eps),
            "csi": hits / (hits + misses + fas + eps), "bias": (bias / np.float32(np.log(np.float32(2.0)))) ** 2}
