"""CPU restatement of the per-case evaluation metrics of the valid / test drivers (/root/reference/U/valid.py:86-111).

TEST INFRASTRUCTURE ONLY (tests/): the product path is cfd_amd.ops.eval_metrics on the GPU.

The reference takes scikit-learn's `cosine_similarity([y[i]], [Y[i]])` per row (:110); scikit-learn is installed in the
build image, so tests/test_cpu_oracle_and_abi.py pins `row_cosine` below against it.
"""
import numpy as np


def row_cosine(a: np.ndarray, b: np.ndarray) -> float:
    """sklearn.metrics.pairwise.cosine_similarity([a], [b])[0][0]: rows are L2-normalised (a zero row stays zero)"""
    na, nb = np.sqrt(np.dot(a, a)), np.sqrt(np.dot(b, b))
    na = na if na > 0 else 1.0
    nb = nb if nb > 0 else 1.0
    return float(np.dot(a / na, b / nb))


def field_metrics(y: np.ndarray, Y: np.ndarray):
    """y (prediction), Y (CFD): (H, W) float32 -> (err_max, err_pct_at_argmax, err_mean, cos_mean, colmax (W,))"""
    y_error = abs(Y - y)                                                   # :86
    err_avg = np.sum(y_error) / y_error.size                               # :96-97 (64*512 there)
    colmax = np.asarray(y_error.max(axis=0))                               # :100
    err_max = colmax.max()                                                 # :101
    mi = np.unravel_index(np.argmax(y_error), y_error.shape)               # :103-104
    with np.errstate(divide="ignore", invalid="ignore"):
        pct = (y_error[mi] / Y[mi]) * 100                                  # :105
    cos = [row_cosine(y[i], Y[i]) for i in range(len(y)) if not np.all(Y[i] == 0)]   # :107-110
    cos_mean = float(np.mean(cos)) if cos else float("nan")                # :111
    return float(err_max), float(pct), float(err_avg), cos_mean, colmax
