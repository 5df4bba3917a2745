from t2ms_amd.model.pretrained.vqvae import Residual, ResidualStack, Encoder, Decoder, vqvae  # noqa: F401
