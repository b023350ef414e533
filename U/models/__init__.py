"""`from models import *` surface of the reference's U/models (U/models/__init__.py:1-3), on cfd_amd."""
from cfd_amd.keras_api import (BatchNormalization, MC_huber_loss, MC_mae, MC_mse, define_discriminator, define_gan,  # noqa: F401
                               load_model, save_model)
from cfd_amd.keras_api import define_generator_U as define_generator  # noqa: F401
