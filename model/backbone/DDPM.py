from t2ms_amd.model.backbone.DDPM import *  # noqa: F401,F403
from t2ms_amd.model.backbone.DDPM import DDPM, gather  # noqa: F401
