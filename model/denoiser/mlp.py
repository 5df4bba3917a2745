from t2ms_amd.model.denoiser.mlp import MLP, MLPlayer, TextToSeriesCrossAttention, TimeEmbedding  # noqa: F401
