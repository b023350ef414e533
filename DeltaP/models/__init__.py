"""`from models import *` surface of the reference's DeltaP/models, on cfd_amd."""
from cfd_amd.keras_api import define_cnn_model, load_model, save_model  # noqa: F401
