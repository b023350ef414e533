"""`from models import *` surface of the reference's P/models, on cfd_amd."""
from cfd_amd.keras_api import (BatchNormalization, MC_huber_loss, MC_mae, MC_mse, define_discriminator, define_gan,  # noqa: F401
                               load_model, save_model)
from cfd_amd.keras_api import define_generator_P as define_generator  # noqa: F401
