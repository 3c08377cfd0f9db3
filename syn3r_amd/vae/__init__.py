from .model import AutoencoderKLTemporalDecoder, DiagonalGaussianDistribution  # noqa: F401
