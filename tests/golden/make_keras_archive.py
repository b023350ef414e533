#!/usr/bin/env python
"""Hand-written `.keras` archives for tests/test_boundary_gpu.py::test_load_model_reads_hand_laid_out_archive.

    /opt/conda/bin/python3.9 tests/golden/make_keras_archive.py cnn|disc <out.keras> <H> <W>

INDEPENDENT of cfd_amd/keras_h5.py on purpose (it imports nothing from the repository and runs under the image's conda
interpreter, the one that has h5py): every HDF5 path below is typed out literally from Keras 3's saving rule
(`keras/src/saving/saving_lib.py`: `layers/<snake_case(class)>[_<k>]/vars/<i>` with <k> counting earlier layers of the same
class in `model.layers` order, `optimizer/vars/<i>` = iterations, learning_rate, then momentum and velocity per trainable
variable in turn) applied by hand to the reference's two smallest graphs, DeltaP/models/generator.py:15-49 and
U/models/discriminator.py:12-44.  A wrong variable order in keras_h5.weight_paths / trainable_order / split_optimizer
fails the test that loads these; the archives are generated at test time (tens of MB: not committed).  STILL UNPINNED
AGAINST REAL KERAS: nobody here could run Keras, so this pins the loader to the published rule as read by a second pair of
eyes, not to Keras' own output.

Variable contents are a formula of (variable id, element index) that the test restates, so that it can tell WHICH dataset
landed in which tensor.
"""
import json
import os
import sys
import tempfile
import zipfile

import h5py
import numpy as np


def fill(var_id, shape, positive=False):
    n = int(np.prod(shape)) if len(shape) else 1
    v = ((np.arange(n, dtype=np.int64) * 7919 + var_id * 104729) % 2003).astype(np.float32) / 2003.0 - 0.5
    v = v * 0.05
    if positive:
        v = np.abs(v) + 0.01
    return v.reshape(shape)


def cnn_layout(H, W):
    """DeltaP/models/generator.py:15-49: Input, Conv2D, LeakyReLU, MaxPooling2D, Conv2D, LeakyReLU, Conv2D, MaxPooling2D,
    Conv2D, MaxPooling2D, Flatten, Dense, Dropout, Dense, Dense"""
    flat = (H // 8) * (W // 8) * 512
    weights = [      # (path, Keras shape)            variable id = position in this list
        ("layers/conv2d/vars/0", (3, 3, 1, 64)), ("layers/conv2d/vars/1", (64,)),
        ("layers/conv2d_1/vars/0", (3, 3, 64, 128)), ("layers/conv2d_1/vars/1", (128,)),
        ("layers/conv2d_2/vars/0", (3, 3, 128, 256)), ("layers/conv2d_2/vars/1", (256,)),
        ("layers/conv2d_3/vars/0", (3, 3, 256, 512)), ("layers/conv2d_3/vars/1", (512,)),
        ("layers/dense/vars/0", (flat, 256)), ("layers/dense/vars/1", (256,)),
        ("layers/dense_1/vars/0", (256, 128)), ("layers/dense_1/vars/1", (128,)),
        ("layers/dense_2/vars/0", (128, 1)), ("layers/dense_2/vars/1", (1,)),
    ]
    empty = ["vars", "layers/input_layer/vars", "layers/leaky_re_lu/vars", "layers/leaky_re_lu_1/vars",
             "layers/max_pooling2d/vars", "layers/max_pooling2d_1/vars", "layers/max_pooling2d_2/vars",
             "layers/flatten/vars", "layers/dropout/vars"]
    # model.trainable_variables: every variable above, in that order (no BatchNorm in this graph)
    trainable = list(range(len(weights)))
    return weights, empty, trainable


def disc_layout(H, W):
    """U/models/discriminator.py:12-44: 2 x Input, Concatenate, Conv2D, LeakyReLU, (Conv2D, BatchNormalization,
    LeakyReLU) x 4, Conv2D, Activation"""
    weights = [
        ("layers/conv2d/vars/0", (4, 4, 2, 128)), ("layers/conv2d/vars/1", (128,)),
        ("layers/conv2d_1/vars/0", (4, 4, 128, 256)), ("layers/conv2d_1/vars/1", (256,)),
        ("layers/batch_normalization/vars/0", (256,)), ("layers/batch_normalization/vars/1", (256,)),        # gamma, beta
        ("layers/batch_normalization/vars/2", (256,)), ("layers/batch_normalization/vars/3", (256,)),        # moving mean, variance
        ("layers/conv2d_2/vars/0", (4, 4, 256, 512)), ("layers/conv2d_2/vars/1", (512,)),
        ("layers/batch_normalization_1/vars/0", (512,)), ("layers/batch_normalization_1/vars/1", (512,)),
        ("layers/batch_normalization_1/vars/2", (512,)), ("layers/batch_normalization_1/vars/3", (512,)),
        ("layers/conv2d_3/vars/0", (4, 4, 512, 1024)), ("layers/conv2d_3/vars/1", (1024,)),
        ("layers/batch_normalization_2/vars/0", (1024,)), ("layers/batch_normalization_2/vars/1", (1024,)),
        ("layers/batch_normalization_2/vars/2", (1024,)), ("layers/batch_normalization_2/vars/3", (1024,)),
        ("layers/conv2d_4/vars/0", (4, 4, 1024, 1024)), ("layers/conv2d_4/vars/1", (1024,)),
        ("layers/batch_normalization_3/vars/0", (1024,)), ("layers/batch_normalization_3/vars/1", (1024,)),
        ("layers/batch_normalization_3/vars/2", (1024,)), ("layers/batch_normalization_3/vars/3", (1024,)),
        ("layers/conv2d_5/vars/0", (4, 4, 1024, 1)), ("layers/conv2d_5/vars/1", (1,)),
    ]
    empty = ["vars", "layers/input_layer/vars", "layers/input_layer_1/vars", "layers/concatenate/vars",
             "layers/leaky_re_lu/vars", "layers/leaky_re_lu_1/vars", "layers/leaky_re_lu_2/vars",
             "layers/leaky_re_lu_3/vars", "layers/leaky_re_lu_4/vars", "layers/activation/vars"]
    # trainable_variables: kernel, bias, then the BatchNorm's gamma, beta -- moving statistics are not trainable
    trainable = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 14, 15, 16, 17, 20, 21, 22, 23, 26, 27]
    return weights, empty, trainable


def main():
    kind, out, H, W = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    weights, empty, trainable = {"cnn": cnn_layout, "disc": disc_layout}[kind](H, W)
    lr, iterations = {"cnn": (1e-3, 5), "disc": (2e-3, 3)}[kind]
    tmp = tempfile.mkdtemp()
    hp = os.path.join(tmp, "model.weights.h5")
    with h5py.File(hp, "w") as f:
        for g in empty:
            f.require_group(g)
        for vid, (path, shape) in enumerate(weights):
            positive = path.startswith("layers/batch_normalization") and path.endswith("/3")     # a variance
            f[path] = fill(vid, shape, positive)
        f["optimizer/vars/0"] = np.int64(iterations)
        f["optimizer/vars/1"] = np.float32(lr)
        for k, vid in enumerate(trainable):
            shape = weights[vid][1]
            f["optimizer/vars/%d" % (2 + 2 * k)] = fill(1000 + vid, shape)                 # momentum of variable vid
            f["optimizer/vars/%d" % (3 + 2 * k)] = fill(2000 + vid, shape, positive=True)  # velocity of variable vid
    inputs = [{"module": "keras.layers", "class_name": "InputLayer", "registered_name": None,
               "config": {"batch_shape": [None, H, W, 1], "dtype": "float32", "sparse": False, "name": "input_layer"}}]
    cfg = {"module": "keras", "class_name": "Functional", "registered_name": "Functional",
           "config": {"name": "functional", "trainable": True, "layers": inputs}}
    with zipfile.ZipFile(out, "w") as z:
        z.writestr("metadata.json", json.dumps({"keras_version": "3.3.3", "date_saved": "2024-06-01@00:00:00"}))
        z.writestr("config.json", json.dumps(cfg))
        z.write(hp, "model.weights.h5")
    os.remove(hp)
    os.rmdir(tmp)
    print(json.dumps({"variables": len(weights), "trainable": len(trainable)}))


if __name__ == "__main__":
    main()
