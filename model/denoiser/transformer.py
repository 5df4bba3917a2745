from t2ms_amd.model.denoiser.transformer import *  # noqa: F401,F403
from t2ms_amd.model.denoiser.transformer import _Attention, _Mlp  # noqa: F401
