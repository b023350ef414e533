"""Helpers the reference's scripts import from `utils` (/root/reference/utils.py), on cfd_amd:
str2bool (:23-29), generate_real_samples (:33-49), generate_fake_samples (:52-62), load_data (:64-107)."""
import numpy as np

from cfd_amd.cli import str2bool  # noqa: F401
from cfd_amd.data import load_data, signed_distance  # noqa: F401


def generate_real_samples(x_train_1, y_train_2, ix, patch_shape):
    X1 = np.asarray([x_train_1[i] for i in ix]).astype("float32")
    X2 = np.asarray([y_train_2[i] for i in ix]).astype("float32")
    y = np.ones((len(ix), patch_shape, patch_shape * 8, 1))
    return [X1, X2], y


def generate_fake_samples(g_model, X_realA, patch_shape):
    """one batched prediction (per-sample BatchNorm keeps it equal to the reference's one-predict-per-case loop)"""
    X = np.asarray(g_model.predict([np.asarray(X_realA)])).astype("float32")
    y = np.zeros((len(X), patch_shape, patch_shape * 8, 1))
    return X, y
