"""Every Keras/TensorFlow default the oracle (and therefore the HIP path) relies on.

Restated from Keras-3 / TF-2.16+ behaviour (the versions named in the reference's
comments, ``U/train.py:145-147``); UNVERIFIED in this container because TensorFlow
is not installed.  One table, so a session that can run Keras fixes one place.
"""

# BatchNormalization() defaults (U/models/generator.py:21,33; discriminator.py:26)
BN_EPSILON = 1e-3          # keras.layers.BatchNormalization(epsilon=0.001)
BN_MOMENTUM = 0.99         # moving = moving*momentum + batch*(1-momentum)
BN_BIASED_VARIANCE = True  # tf.nn.moments -> population variance, also for the moving average
BN_GAMMA_INIT = 1.0
BN_BETA_INIT = 0.0
BN_MOVING_MEAN_INIT = 0.0
BN_MOVING_VAR_INIT = 1.0

# LeakyReLU(alpha=0.2) / LeakyReLU(negative_slope=0.2) (U/models/generator.py:23, P/...:48)
LEAKY_SLOPE = 0.2

# Dropout(0.5)(g, training=True): inverted dropout, kept values scaled by 1/(1-rate)
DROPOUT_RATE = 0.5
DROPOUT_KEEP_SCALE = 2.0

# RandomNormal(stddev=0.02) kernels, zero biases
INIT_STDDEV = 0.02

# Kernel layouts
#   Conv2D.kernel           (kh, kw, Cin,  Cout)
#   Conv2DTranspose.kernel  (kh, kw, Cout, Cin)
#   Dense.kernel            (in, out)
# Flatten() keeps NHWC order: index = (h*W + w)*C + c

# padding='same' (TF rule): out = ceil(in/stride);
#   total = max((out-1)*stride + k - in, 0); before = total//2; after = total-before
#   k4 s2 even in -> (1,1);  k4 s1 -> (1,2);  k3 s1 -> (1,1)
# Conv2DTranspose k4 s2 'same': out = 2*in, out[2*i - 1 + k] += x[i]*W[k]
#   == torch.conv_transpose2d(stride=2, padding=1), no kernel flip.

# Adam (keras.optimizers.Adam): beta_2=0.999, epsilon=1e-7, epsilon OUTSIDE the
# bias-corrected sqrt:
#   alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2)
#   w -= alpha*m/(sqrt(v)+eps)
ADAM_BETA_2 = 0.999
ADAM_EPSILON = 1e-7

# binary_crossentropy on a model whose last op is Activation('sigmoid'):
# Keras-3 caches the logits (`output._keras_logits`) and evaluates
# tf.nn.sigmoid_cross_entropy_with_logits = max(x,0) - x*z + log1p(exp(-|x|)).
# The probability form clips p to [1e-7, 1-1e-7] first.  The two agree to fp32
# rounding except at saturation; BCE_FROM_LOGITS selects which one is restated.
BCE_FROM_LOGITS = True
BCE_CLIP = 1e-7

# Losses reduce with a plain mean over every element (sum_over_batch_size);
# compiled loss = sum_i loss_weights[i]*loss_i.
# Huber(delta=1.0): where(|e|<=d, 0.5 e^2, d|e| - 0.5 d^2)

# define_gan freezes every non-BN layer of D (U/models/generator.py:79-81).
# Keras-2 / pix2pix intent: D still trains in its own train_on_batch (it was
# compiled before the freeze) and only its BN gamma/beta train inside the GAN.
# Under Keras 3 the flag may be read at trace time and freeze D's convs in D's
# own step as well -- open question, selectable by this flag.
D_CONVS_TRAIN_IN_D_STEP = True
