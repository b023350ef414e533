"""Drivers behind the U|P|DeltaP/{train,valid,test}.py entry points: the reference's flags, file names and
step order (U/train.py, P/train.py, DeltaP/train.py, */valid.py, */test.py), with the model calls going
to cfd_amd.keras_api.  Plots are written only when matplotlib imports (host-side reporting is outside the
hot path)."""
from __future__ import annotations

import argparse
import csv
import os
import random
import time
from typing import Optional

import numpy as np

from . import data as D
from . import keras_api as K

FIELDS = {
    "U": dict(target="u", unit="m/s", default_dir="U/training_U/", root="./U/", label="UX", gan=True),
    "P": dict(target="p", unit="Pa", default_dir="P/training_P/", root="./P/", label="P", gan=True),
    "DeltaP": dict(target="dp", unit="Pa", default_dir="DeltaP/training_DeltaP/", root="./DeltaP/", label="P", gan=False),
}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("true",):
        return True
    if v.lower() in ("false",):
        return False
    return None


def train_parser(field: str) -> argparse.ArgumentParser:
    f = FIELDS[field]
    ap = argparse.ArgumentParser(description="Input option for training GAN CFD model")
    ap.add_argument("-c", "--checkpoint", default=False, type=str2bool, help="Whether load checkpoint for training")
    ap.add_argument("-e", "--epocheval", default=1, type=int, help="Number of epoch for each evaluation")
    ap.add_argument("-n", "--numepoch", default=10000, type=int, help="Number of epoch for training iteration")
    ap.add_argument("-t", "--trainpath", default="30_case_duyanh/processed/", type=str, help="training directory input")
    ap.add_argument("-d", "--directory", default=f["default_dir"], type=str, help="training directory output")
    return ap


def _nhwc(arrs):
    return np.expand_dims(np.asarray(arrs).astype("float32"), axis=3)


def _split(n):
    from sklearn.model_selection import train_test_split
    return train_test_split(range(n), test_size=0.1, random_state=42)


def _resume_step(dir_name, train_size, chkpt):
    if not chkpt:
        return 0
    try:
        with open(dir_name + "steps.txt") as f:
            return (int(f.read().strip()) // train_size) * train_size
    except FileNotFoundError:
        return 0


def _plot_losses(dir_name, curves):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return
    for fname, ylabel, series in curves:
        plt.figure()
        for vals, style, label in series:
            plt.plot(range(len(vals)), vals, style, label=label)
        plt.legend(loc="upper center")
        plt.xlabel("Epoch number")
        plt.ylabel(ylabel)
        plt.savefig(dir_name + fname)
    plt.close("all")


def train(field: str, argv=None, image_shape=(64, 512, 1), models=None):
    """One process = the reference's training script.  `models` lets tests inject small pre-built models."""
    args = train_parser(field).parse_args(argv)
    f = FIELDS[field]
    dir_name = f["root"] + os.path.basename(os.path.normpath(f["default_dir"])) + "/"   # hard-coded in the reference
    ckpt_dir = dir_name + "model_ckpt/"
    os.makedirs(dir_name, exist_ok=True)

    if models is not None:
        d_model, g_model, gan_model = models
    elif args.checkpoint:
        g_model = K.load_model(ckpt_dir + "model_G.keras")
        if f["gan"]:
            d_model = K.load_model(ckpt_dir + "model_D.keras")
            gan_model = K.link_gan(K.load_model(ckpt_dir + "model_GAN.keras"), g_model, d_model)
        else:
            d_model = gan_model = None
    else:
        if f["gan"]:
            d_model = K.define_discriminator(image_shape)
            g_model = K.define_generator_U(image_shape) if field == "U" else K.define_generator_P(image_shape)
            gan_model = K.define_gan(g_model, d_model, image_shape)
        else:
            d_model = gan_model = None
            g_model = K.define_cnn_model(image_shape)

    bnd, sp, su, dps = D.load_data(args.trainpath, image_shape[0], image_shape[1])
    train_idx, valid_idx = _split(len(bnd))
    np.savetxt(dir_name + "train_indices.txt", train_idx, fmt="%d")
    np.savetxt(dir_name + "valid_indices.txt", valid_idx, fmt="%d")
    x_train, x_valid = _nhwc([bnd[i] for i in train_idx]), _nhwc([bnd[i] for i in valid_idx])
    if field == "U":
        y_train, y_valid = _nhwc([su[i] for i in train_idx]), _nhwc([su[i] for i in valid_idx])
        scale = 1.0
    elif field == "P":
        y_train, y_valid = _nhwc([sp[i] for i in train_idx]), _nhwc([sp[i] for i in valid_idx])
        scale = float(y_train.max())                       # P/train.py:79-82
        y_train, y_valid = y_train / scale, y_valid / scale
    else:
        y_train = np.asarray([dps[i] for i in train_idx], dtype="float32")[:, None]
        y_valid = np.asarray([dps[i] for i in valid_idx], dtype="float32")[:, None]
        scale = float(y_train.max())                       # DeltaP/train.py:83-85
        y_train, y_valid = y_train / scale, y_valid / scale
    print("x1 train shape", x_train.shape)
    print(f"y shape: {y_train.shape}, min = {y_train.min()}, max = {y_train.max()}")
    print("x1 valid shape", x_valid.shape)

    train_size, valid_size = len(x_train), len(x_valid)
    n_steps = train_size * args.numepoch
    start = _resume_step(dir_name, train_size, args.checkpoint)
    pool = list(range(train_size))
    sums = [0.0, 0.0, 0.0]
    hist = dict(d1=[], d2=[], g=[], val=[])
    min_eval = -1.0
    n_patch = d_model.output_shape[1] if f["gan"] else 1
    t0 = time.time()
    for i in range(start, n_steps):
        k = random.choice(pool)                            # batch 1, sampling without replacement per epoch
        pool.remove(k)
        A, B = x_train[[k]], y_train[[k]]
        if f["gan"]:
            y_real = np.ones((1, n_patch, n_patch * 8, 1))
            y_fake = np.zeros((1, n_patch, n_patch * 8, 1))
            fake = g_model.predict([A])
            d1 = d_model.train_on_batch([A, B], y_real)
            d2 = d_model.train_on_batch([A, fake], y_fake)
            g_loss, _, _ = gan_model.train_on_batch([A], [y_real, B])
            print(">%d, d1[%.3f] d2[%.3f] g[%.3f]" % (i + 1, d1, d2, g_loss))
            sums = [sums[0] + d1, sums[1] + d2, sums[2] + g_loss]
        else:
            g_loss = g_model.train_on_batch([A], [B])
            print(">%d, g[%.3f]" % (i + 1, g_loss))
            sums[2] += g_loss
        with open(dir_name + "steps.txt", "w") as fh:
            fh.write(str(i + 1))
        if (i % train_size) + 1 == train_size:              # epoch end
            print("epoch time: {}".format(time.time() - t0))
            pool = list(range(train_size))
            val = 0.0
            for v in range(valid_size):
                if f["gan"]:
                    val += gan_model.test_on_batch([x_valid[[v]]], [np.ones((1, n_patch, n_patch * 8, 1)), y_valid[[v]]])[0]
                else:
                    val += g_model.test_on_batch([x_valid[[v]]], [y_valid[[v]]])
            val /= max(valid_size, 1)
            if field == "U" and (min_eval == -1 or val < min_eval):
                min_eval = val
                K.save_model(g_model, dir_name + "model_eval.keras", overwrite=True, include_optimizer=True)
                print("---->Saved model: %s" % (dir_name + "model_eval.keras"))
            hist["d1"].append(sums[0] / train_size)
            hist["d2"].append(sums[1] / train_size)
            hist["g"].append(sums[2] / train_size)
            hist["val"].append(val)
            sums = [0.0, 0.0, 0.0]
            curves = [("training_G_loss.png", "Training loss", [(hist["g"], "-b", "G loss")]),
                      ("validation_loss.png", "Validation loss", [(hist["val"], "-y", "Val loss")])]
            if f["gan"]:
                curves.insert(0, ("training_D_loss.png", "Training loss",
                                  [(hist["d1"], "-r", "D1 loss"), (hist["d2"], "-g", "D2 loss")]))
            _plot_losses(dir_name, curves)
            os.makedirs(ckpt_dir, exist_ok=True)
            K.save_model(g_model, ckpt_dir + "model_G.keras", overwrite=True, include_optimizer=True)
            if f["gan"]:
                K.save_model(d_model, ckpt_dir + "model_D.keras", overwrite=True, include_optimizer=True)
                K.save_model(gan_model, ckpt_dir + "model_GAN.keras", overwrite=True, include_optimizer=True)
            print("---->Saved checkpoint: %s" % (ckpt_dir + "model_G.keras"))
            t0 = time.time()
    return hist


# What each of the reference's six evaluation scripts scores, and with which de-normalisation (file:line under the
# reference root).  `select`: "tail" = the last 10 % of the folder in listing order; "all" = every case of the folder;
# "indices" = the cases of valid_indices.txt saved by the training script.  `scale_from`: where P_max comes from --
# "folder" = the train_indices.txt cases of the evaluated folder, "train_folder" = those of a second (training) folder.
# `factor`: the multiple of P_max the prediction is multiplied with, AS WRITTEN in the reference (SURVEY.md "quirks":
# P trains on y / P_max but evaluates with 2 * P_max; DeltaP/test.py is a copy of P/test.py and inherits its factor).
EVAL_PLANS = {
    ("U", "valid"): dict(folder="./100_case_si/processed/", select="tail", scale_from=None, factor=1.0),           # U/valid.py:41-51
    ("U", "test"): dict(folder="./30_case_duyanh/processed/", select="all", scale_from=None, factor=1.0),          # U/test.py:43,77-86
    ("P", "valid"): dict(folder="./100_case_si_test/processed/", select="indices", scale_from="folder", factor=2.0),  # P/valid.py:42-66,95
    ("P", "test"): dict(folder="./30_case_duyanh/processed/", train_folder="./100_case_si/processed/", select="all",
                        scale_from="train_folder", factor=2.0),                                                    # P/test.py:43,76-90,123-125,168
    ("DeltaP", "valid"): dict(folder="./130_case/processed/", select="indices", scale_from="folder", factor=1.0),  # DeltaP/valid.py:78-126,157
    ("DeltaP", "test"): dict(folder="./30_case_duyanh/processed/", train_folder="./100_case_si/processed/", select="all",
                             scale_from="train_folder", factor=2.0),                                               # DeltaP/test.py:43,76-90,123-125,168
}
EVAL_CSV = "error_test_UX_ibm_cylinder.csv"      # the file name every one of the six scripts writes (U/valid.py:69)


def eval_selection(field: str, mode: str, n_cases: int, train_idx=None, valid_idx=None):
    """-> indices (into the evaluated folder's listing order) of the cases `<field>/<mode>.py` scores.  Pure host logic."""
    plan = EVAL_PLANS[(field, mode)]
    if plan["select"] == "tail":                                  # U/valid.py:44-51
        train_size = int(0.9 * n_cases + 1)
        test_size = n_cases - train_size
        if test_size <= 0:
            raise ValueError("not enough cases for a hold-out split")
        return list(range(n_cases - test_size, n_cases))
    if plan["select"] == "all":                                   # U/test.py:77-86 (train_size = 0)
        return list(range(n_cases))
    if valid_idx is None:
        raise ValueError(f"{field}/{mode}.py selects its cases with valid_indices.txt: none given")
    idx = [int(i) for i in np.atleast_1d(valid_idx)]
    if any(i < 0 or i >= n_cases for i in idx):
        raise ValueError("valid_indices.txt does not belong to this folder (index out of range)")
    return idx


def eval_scale(field: str, mode: str, train_targets, train_idx=None) -> float:
    """-> the number a prediction is multiplied with.  `train_targets`: the per-case targets (P grids, or DeltaP values)
    of the folder `scale_from` names, in listing order; P_max = max over its train_indices.txt cases
    (P/valid.py:53-66, P/test.py:79-84, DeltaP/valid.py:92-124)."""
    plan = EVAL_PLANS[(field, mode)]
    if plan["scale_from"] is None:
        return 1.0
    if train_idx is None:
        raise ValueError(f"{field}/{mode}.py takes P_max from the train_indices.txt cases: none given")
    idx = [int(i) for i in np.atleast_1d(train_idx)]
    if not idx or any(i < 0 or i >= len(train_targets) for i in idx):
        raise ValueError("train_indices.txt does not belong to the training folder (index out of range)")
    sel = np.asarray([train_targets[i] for i in idx])
    if field == "DeltaP" and mode == "valid":
        sel = sel.astype("float32")                               # DeltaP/valid.py:110 casts before :124's max
    return float(plan["factor"]) * float(sel.max())


def evaluate(field: str, processed_folder: Optional[str] = None, model_path: Optional[str] = None,
             out_dir: Optional[str] = None, model=None, batch: int = 32, image_shape=(64, 512, 1), mode: str = "valid",
             train_folder: Optional[str] = None, indices_dir: Optional[str] = None):
    """`<field>/valid.py` (mode "valid") or `<field>/test.py` (mode "test"): which cases are scored and how the
    prediction is de-normalised follow EVAL_PLANS, one entry per reference script; per-case abs / % error, row-wise
    cosine similarity, one CSV row per case (U/valid.py:71-117; DeltaP/valid.py:146-173 writes abs / relative error).
    `indices_dir` holds train_indices.txt / valid_indices.txt (default `<root>/training_<field>/`, where the training
    script saved them).  Predictions are BATCHED here (per-sample BatchNorm makes a batch of B equal to B single
    predictions); the reference loops one predict per case (utils.py:52-62)."""
    f = FIELDS[field]
    plan = EVAL_PLANS[(field, mode)]
    train_dir = f["root"] + os.path.basename(os.path.normpath(f["default_dir"])) + "/"
    processed_folder = processed_folder or plan["folder"]
    model_path = model_path or (train_dir + "model_ckpt/model_G.keras")
    out_dir = out_dir or (f["root"] + ("evaluation_valid/" if mode == "valid" else "evaluation_test/"))
    indices_dir = indices_dir or train_dir
    model = model or K.load_model(model_path, compile=False)
    bnd, sp, su, dps = D.load_data(processed_folder, image_shape[0], image_shape[1])
    train_idx = valid_idx = None
    if plan["select"] == "indices" or plan["scale_from"]:
        train_idx = np.loadtxt(os.path.join(indices_dir, "train_indices.txt"), dtype=int)
    if plan["select"] == "indices":
        valid_idx = np.loadtxt(os.path.join(indices_dir, "valid_indices.txt"), dtype=int)
    sel = eval_selection(field, mode, len(bnd), train_idx, valid_idx)
    if plan["scale_from"] == "train_folder":
        _b, tsp, _su, tdps = D.load_data(train_folder or plan["train_folder"], image_shape[0], image_shape[1])
        scale = eval_scale(field, mode, tsp if field == "P" else tdps, train_idx)
    else:
        scale = eval_scale(field, mode, sp if field == "P" else dps, train_idx)
    test_size = len(sel)
    x_test = _nhwc([bnd[i] for i in sel])
    if field == "U":
        y_test = np.asarray([su[i] for i in sel], dtype="float32")
    elif field == "P":
        y_test = np.asarray([sp[i] for i in sel], dtype="float32")
    else:
        y_test = np.asarray([dps[i] for i in sel], dtype="float32")
    os.makedirs(out_dir, exist_ok=True)
    rows, mean_err = [], 0.0
    if field == "DeltaP":
        scale32 = np.float32(scale)
        pred = np.concatenate([model.predict(x_test[s:s + batch]) for s in range(0, test_size, batch)], axis=0) * scale32
        y_test = (y_test / scale32) * scale32                     # DeltaP/valid.py:126,159: normalised, then scaled back, in fp32
    else:
        # fields stay on the device: one metrics launch per batch, four numbers per case come back (U/valid.py:86-111)
        import torch
        from . import ops
        metrics = []
        for s in range(0, test_size, batch):
            y_dev = model.predict_on_device(x_test[s:s + batch])
            t_dev = torch.from_numpy(np.ascontiguousarray(y_test[s:s + batch], dtype=np.float32)).to(y_dev.device)
            m, _colmax = ops.eval_metrics(y_dev, t_dev, scale)
            metrics.append(m)
        metrics = np.concatenate(metrics, axis=0)
    csv_path = os.path.join(out_dir, EVAL_CSV)
    with open(csv_path, "w", newline="") as fh:
        w = csv.writer(fh)
        if field == "DeltaP":
            w.writerow(["Test ID", "Abs Error (Pa)", "Relative (%)"])                       # DeltaP/valid.py:148
        else:
            w.writerow(["Test ID", f"Error max ({f['unit']})", "Error max (%)", f"Error mean ({f['unit']})", "Cosine Similarity"])
        for ix in range(test_size):
            if field == "DeltaP":
                yv, Yv = np.float32(pred[ix, 0]), np.float32(y_test[ix])
                err = abs(Yv - yv)
                with np.errstate(divide="ignore", invalid="ignore"):
                    row = [ix, float(err), float(err / Yv * np.float32(100))]               # :170-173
                mean_err += float(err)
            else:
                row = [ix] + [float(v) for v in metrics[ix]]
                mean_err += row[3]
            w.writerow([str(v) for v in row])
            rows.append(row)
    mean_err /= max(test_size, 1)
    print("Total average error: ", mean_err)
    return rows, mean_err
